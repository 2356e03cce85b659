"""TEST INFRASTRUCTURE ONLY (oracle).  ctypes front-end of oracle/env.c: a sequential CPU environment with the
reference's reset/step semantics over explicit tables."""
import ctypes
import numpy as np
from . import build as _b

OBS_DIM, OBS_LD = 779, 780
c_p = ctypes.c_void_p


class Tables(ctypes.Structure):
    _fields_ = [('size', c_p), ('quality', c_p), ('video_len', c_p), ('n_chunk_max', ctypes.c_int),
                ('vp_gt', c_p), ('vp_pred', c_p), ('vp_acc', c_p), ('vp_start', c_p), ('vp_end', c_p), ('n_vpchunk_max', ctypes.c_int),
                ('trace_bw', c_p), ('trace_len', c_p), ('trace_len_max', ctypes.c_int),
                ('samples', c_p), ('n_sample', ctypes.c_int), ('qoe_w', c_p),
                ('video_rates', ctypes.c_int * 5), ('startup_download', ctypes.c_int), ('chunk_length', ctypes.c_int),
                ('max_size', ctypes.c_double), ('max_throughput', ctypes.c_double), ('train_identifier_reward', ctypes.c_int)]


class EnvTables:
    """Holds numpy arrays (kept alive) + the C struct."""

    def __init__(self, arrays, qoe_w, video_rates=(1, 5, 8, 16, 35), startup_download=5, chunk_length=1, max_size=500000,
                 max_throughput=5000000, train_identifier_reward=False):
        a = {k: np.ascontiguousarray(v) for k, v in arrays.items()}
        a['size'] = a['size'].astype(np.int32)
        a['quality'] = a['quality'].astype(np.float32)
        a['qoe_w'] = np.ascontiguousarray(qoe_w, dtype=np.float32)
        self.a = a
        t = Tables()
        for k in ('size', 'quality', 'video_len', 'vp_gt', 'vp_pred', 'vp_acc', 'vp_start', 'vp_end', 'trace_bw', 'trace_len', 'samples',
                  'qoe_w'):
            setattr(t, k, a[k].ctypes.data)
        t.n_chunk_max = a['size'].shape[1]
        t.n_vpchunk_max = a['vp_gt'].shape[1]
        t.trace_len_max = a['trace_bw'].shape[1]
        t.n_sample = a['samples'].shape[0]
        for i, r in enumerate(video_rates):
            t.video_rates[i] = int(r)
        t.startup_download, t.chunk_length = startup_download, chunk_length
        t.max_size, t.max_throughput = float(max_size), float(max_throughput)
        t.train_identifier_reward = int(train_identifier_reward)
        self.c = t


class Env:
    def __init__(self, tables, seed=0, worker_num=1):
        L = _b.lib()
        assert L.oracle_env_tables_size() == ctypes.sizeof(Tables)
        self.L, self.T = L, tables
        self.state = ctypes.create_string_buffer(L.oracle_env_state_size())
        ctypes.memset(self.state, 0, len(self.state))
        ints = ctypes.cast(self.state, ctypes.POINTER(ctypes.c_int))
        ints[0] = seed % worker_num      # worker_id
        ints[1] = worker_num
        self.obs = np.zeros(OBS_LD, np.float32)

    @property
    def sample_id(self):
        return ctypes.cast(self.state, ctypes.POINTER(ctypes.c_int))[2]

    def reset(self):
        self.L.oracle_env_reset(ctypes.byref(self.T.c), self.state, self.obs.ctypes.data_as(c_p))
        return self.obs[:OBS_DIM].copy()

    def step(self, action):
        r = ctypes.c_float()
        parts = (ctypes.c_float * 4)()
        done = self.L.oracle_env_step(ctypes.byref(self.T.c), self.state, int(action), self.obs.ctypes.data_as(c_p), ctypes.byref(r), parts)
        return self.obs[:OBS_DIM].copy(), np.float32(r.value), bool(done), np.array(parts, np.float32)


def allocate_tile_rates(rate_in, rate_out, pred_viewport, video_rates=(1, 5, 8, 16, 35)):
    L = _b.lib()
    pv = np.ascontiguousarray(pred_viewport, dtype=np.float32)
    rates = (ctypes.c_int * 5)(*video_rates)
    out = np.zeros(64, np.int32)
    L.oracle_allocate_tile_rates(int(rate_in), int(rate_out), pv.ctypes.data_as(c_p), rates, out.ctypes.data_as(c_p))
    return out


ACTION2RATES = [(1, 0), (2, 0), (3, 0), (4, 0), (2, 1), (3, 1), (4, 1), (3, 2), (4, 2), (4, 3), (0, 0), (1, 1), (2, 2), (3, 3), (4, 4)]


def generate_environment_samples(nv, nu, nt, nq):
    """utils/common.py:60-84 (zipped cyclic enumeration)."""
    import math
    max_len = max(nv, nu, nt, nq)
    total = max(max_len, nv * nq * math.ceil(max_len / (nv * nq)))
    return np.array([(i % nv, i % nu, i % nt, i % nq) for i in range(total)], np.int32)


def generate_environment_test_samples(nv, nu, nt, nq):
    """utils/common.py:87-98 (full product)."""
    return np.array([(i, j, k, l) for i in range(nv) for j in range(nu) for k in range(nt) for l in range(nq)], np.int32)


class Expert:
    """MPC expert over an `Env` (expert_env.py): profile cache + exhaustive look-ahead `choose_action`."""

    def __init__(self, tables, vp_video, horizon):
        self.L, self.T, self.horizon = _b.lib(), tables, int(horizon)
        self.vp_video = np.ascontiguousarray(vp_video, dtype=np.int32)
        n_vp, nvc = tables.a['vp_gt'].shape[:2]
        self.cache = {k: np.zeros((n_vp, nvc, 15), np.int64 if 'size' in k else np.float32)
                      for k in ('gt_quality', 'pred_quality', 'gt_var', 'pred_var', 'gt_size', 'pred_size')}
        c = self.cache
        self.L.oracle_expert_profile(ctypes.byref(tables.c), self.vp_video.ctypes.data_as(c_p), n_vp, *[
            c[k].ctypes.data_as(c_p) for k in ('gt_quality', 'pred_quality', 'gt_var', 'pred_var', 'gt_size', 'pred_size')])

    def choose_action(self, env, with_value=False):
        c = self.cache
        val, idx = ctypes.c_float(), ctypes.c_longlong()
        a = self.L.oracle_expert_choose(ctypes.byref(self.T.c), env.state, self.horizon, c['pred_quality'].ctypes.data_as(c_p),
                                        c['pred_var'].ctypes.data_as(c_p), c['pred_size'].ctypes.data_as(c_p), ctypes.byref(val),
                                        ctypes.byref(idx))
        return (int(a), np.float32(val.value), int(idx.value)) if with_value else int(a)
