"""CPU-only: the C-ABI library builds for gfx950, loads, and exports every symbol that
include/mansy_hip.h declares; the ctypes table lists the same set; host-side argument validation works
without a GPU (no compute calls here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def built():
    from mansy_immersivevideostreaming_amd import build_ext
    return build_ext.build()


def _declared():
    src = open(os.path.join(ROOT, 'include', 'mansy_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(mansy_[a-z0-9_]+)\s*\(', src)))


def test_header_symbols_exported(built):
    L = ctypes.CDLL(built)
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), n


def test_release_library_has_no_process_wide_state_entry_points(built):
    """ABI 8 ("no global state except an opaque ctx"): no setter, no hook registration, no kernel-selection knob is exported,
    and the sources read no environment variable."""
    import subprocess
    out = subprocess.run(['nm', '-D', '--defined-only', built], check=True, capture_output=True, text=True).stdout
    names = sorted({l.split()[-1] for l in out.splitlines() if ' T ' in l and 'mansy_' in l.split()[-1]})
    assert len(names) >= 90
    bad = [n for n in names if n != 'mansy_set_error' and      # (the thread-local error string's internal writer)
           re.search(r'^mansy_(set_|get_gemm|lab_)|_variant$|mansy_gemm_col_group|mansy_gemm_f32_wsk|bn_sync_hook', n)]
    assert bad == ['mansy_xg_set_timeout_ms'] or bad == [], bad          # (a per-context setting of an opaque ctx is not process-wide state)
    assert ctypes.CDLL(built).mansy_abi_version() == 9
    csrc = os.path.join(ROOT, 'mansy_immersivevideostreaming_amd', 'csrc')
    for f in os.listdir(csrc):
        if f.endswith(('.hip', '.h')):
            assert 'getenv' not in open(os.path.join(csrc, f)).read(), f


def test_package_reads_no_environment_except_the_launcher_variables():
    """Round 6 hygiene: behaviour switches are constructor arguments / attributes, not environment variables.  The only reads under the package are
    dist.py's launcher variables (RANK / WORLD_SIZE / LOCAL_RANK / LOCAL_WORLD_SIZE, the rendezvous backend and the shared-GPU functional-test switch,
    HSA_ENABLE_IPC_MODE_LEGACY) and build_ext.py's HIPCC (a build tool, not the product path)."""
    pkg = os.path.join(ROOT, 'mansy_immersivevideostreaming_amd')
    hits = []
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith('.py'):
                path = os.path.join(d, f)
                for i, line in enumerate(open(path), 1):
                    if re.search(r'os\.environ|getenv', line) and not line.lstrip().startswith('#'):
                        hits.append((os.path.relpath(path, pkg), i))
    assert {h[0] for h in hits} <= {'dist.py', 'build_ext.py'}, hits
    for f in os.listdir(os.path.join(pkg, 'csrc')):            # and no lab conditional in any kernel source (csrc/lab/ is the lab build's own TU)
        if f.endswith('.hip'):
            assert 'MANSY_LAB' not in open(os.path.join(pkg, 'csrc', f)).read(), f


def test_ctypes_table_matches_header(built):
    from mansy_immersivevideostreaming_amd import _lib
    assert _lib.declared_symbols() == _declared()
    _lib.lib()


def test_param_table_matches_reference_keys(built):
    from mansy_immersivevideostreaming_amd import _lib
    from oracle import vp_oracle as vo
    L = _lib.lib()
    for bias, total in ((0, 9189382), (1, 9211910)):      # parameter counts of the two layouts
        cfg = _lib.VPConfig(B=4, S=10, T=10, d_model=512, n_head=8, d_ff=512, n_enc=2, n_dec=2, in_ch=6, has_bias=bias,
                            p_pe=0.2, p_drop=0.1, ln_eps=1e-5, bn_eps=1e-5, bn_momentum=0.1, max_len=5000)
        n = L.mansy_vp_num_params(ctypes.byref(cfg))
        sd = vo.make_state_dict(512, 1, bias=bool(bias))
        ref = [k for k in sd if 'running_' not in k and 'num_batches' not in k and k != 'positional_embedding.pe']
        got, numel_sum = [], 0
        for i in range(n):
            buf = ctypes.create_string_buffer(160)
            numel, nd, shape = ctypes.c_longlong(), ctypes.c_int(), (ctypes.c_longlong * 4)()
            assert L.mansy_vp_param_info(ctypes.byref(cfg), i, buf, 160, ctypes.byref(numel), ctypes.byref(nd), shape) == 0
            k = buf.value.decode()
            assert tuple(sd[k].shape) == tuple(shape[:nd.value]), k
            got.append(k)
            numel_sum += numel.value
        assert got == ref          # same order as the reference state_dict
        assert numel_sum == total


def test_bad_config_is_rejected_with_message(built):
    from mansy_immersivevideostreaming_amd import _lib
    L = _lib.lib()
    cfg = _lib.VPConfig(B=4, S=40, T=10, d_model=512, n_head=8, d_ff=512, n_enc=2, n_dec=2, in_ch=6, has_bias=1,
                        p_pe=0.2, p_drop=0.1, ln_eps=1e-5, bn_eps=1e-5, bn_momentum=0.1, max_len=5000)
    assert L.mansy_vp_num_params(ctypes.byref(cfg)) < 0
    assert b'S' in L.mansy_last_error()
    assert L.mansy_vp_workspace_bytes(ctypes.byref(cfg)) == 0


def test_model_state_dict_layout_cpu(built):
    """Host mirror: key names/shapes/order equal the reference layouts (no GPU needed to construct)."""
    import torch
    from mansy_immersivevideostreaming_amd.viewport_prediction.models import ViewportTransformerMTIO
    from oracle import vp_oracle as vo
    for bias in (True, False):
        m = ViewportTransformerMTIO(in_channel=2, fut_window=10, d_model=64, dim_feedforward=64, device='cpu', bias=bias)
        sd = m.state_dict()
        ref = vo.make_state_dict(64, 3, bias=bias)
        assert list(sd.keys()) == list(ref.keys())
        for k in ref:
            assert tuple(sd[k].shape) == tuple(ref[k].shape), k
        m.load_state_dict(ref)
        for k in ref:
            assert torch.equal(m.state_dict()[k], ref[k]), k
        # loading the other layout switches the model over (SURVEY 8c version trap)
        other = vo.make_state_dict(64, 4, bias=not bias)
        m.load_state_dict(other)
        assert list(m.state_dict().keys()) == list(other.keys())
        with pytest.raises(Exception):
            m.sample(torch.zeros(2, 10, 2), torch.zeros(2, 1, 2))     # CPU tensors: no fallback, must raise


def test_bitrate_selection_host_mirrors_cpu(built):
    """Host mirrors that need no device: parameter tables of the engines equal the reference state_dict layouts (golden key
    lists), observation row <-> dict mappings round-trip, CPU modules refuse to run (no fallback)."""
    import numpy as np
    import torch
    from mansy_immersivevideostreaming_amd._lib import MansyError
    from mansy_immersivevideostreaming_amd.bitrate_selection.envs import expert_env
    from mansy_immersivevideostreaming_amd.bitrate_selection.models import mansy, simple_rl
    here = os.path.dirname(os.path.abspath(__file__))
    Z = np.load(os.path.join(here, 'golden', 'a2c_reference.npz'))
    keys = [str(k) for k in Z['net/keys']]
    table = mansy._Flat(2).table
    assert [n for n, _ in table] == [k for k in keys if not k.startswith('critic.feature_net.')]
    for name, shape in table:
        assert tuple(Z['net/w::' + name].shape) == shape, name
    # A2C observation row <-> SimpleRLEnv state dict
    row = np.arange(416, dtype=np.float32)
    row[395:] = 0
    d = simple_rl.obs_to_dict(row)
    assert d['chunk_sizes'].shape == (5, 64) and d['throughput'].shape == (1, 8) and d['last_bitrates'].shape == (2,)
    back = simple_rl.obs_to_tensor(d, 'cpu').numpy()[0]
    np.testing.assert_array_equal(back, row)
    batched = {k: np.stack([v, v]) for k, v in d.items()}
    assert simple_rl.obs_to_tensor(batched, 'cpu').shape == (2, 416)
    # modules construct on the CPU (state_dict layout) but never compute there
    fn = simple_rl.FeatureNet(8, 64, 5, device='cpu')
    actor, critic = simple_rl.Actor(fn, 640, 15, 'cpu'), simple_rl.Critic(fn, 640, 'cpu')
    pol = simple_rl.A2CPolicy(actor, critic, None, None)
    assert len(pol.state_dict()) == 56
    with pytest.raises(MansyError):
        actor(d)
    # expert helpers
    assert [expert_env.rates2action(*expert_env.action2rates(a)) for a in range(15)] == list(range(15))
    assert expert_env.action2rates(99) == (0, 0) and expert_env.rates2action(0, 4) == 0


def test_lazy_losses_mapping_cpu():
    """update()'s return value: a read-only Mapping that fetches the per-minibatch statistics on first access."""
    import json
    import torch
    from mansy_immersivevideostreaming_amd.bitrate_selection.models.mansy_ppo import LazyLosses, split_indices
    l = LazyLosses(('loss', 'loss/clip'), [torch.tensor([[1., 2.], [3., 4.]]), torch.tensor([[5., 6.]])])
    assert l.n_steps == 3 and len(l) == 2 and list(l) == ['loss', 'loss/clip'] and 'loss' in l and 'x' not in l
    assert l._pending is not None                       # nothing fetched yet
    assert l['loss'] == [1.0, 3.0, 5.0] and l.get('loss/clip') == [2.0, 4.0, 6.0] and l.get('nope', 7) == 7
    assert dict(l) == {'loss': [1.0, 3.0, 5.0], 'loss/clip': [2.0, 4.0, 6.0]} and json.loads(json.dumps(dict(l)))['loss'][2] == 5.0
    empty = LazyLosses(('loss',), [])
    assert empty.n_steps == 0 and empty['loss'] == []
    # tianshou's Batch.split(size, shuffle, merge_last): a short tail is merged into the last chunk
    import numpy as np
    np.random.seed(0)
    chunks = list(split_indices(1100, 512))
    assert [len(c) for c in chunks] == [512, 588] and sorted(np.concatenate(chunks).tolist()) == list(range(1100))
    assert [len(c) for c in split_indices(1024, 512, shuffle=False)] == [512, 512]


def test_ctypes_structs_match_header_layout(tmp_path):
    """Every ctypes.Structure of _lib.py against the struct of include/mansy_hip.h it mirrors: a C program compiled from the
    header prints sizeof and the offset of every field; names, order, offsets and total size must agree."""
    import ctypes
    import re
    import subprocess
    from mansy_immersivevideostreaming_amd import _lib
    pairs = {'mansy_vp_config': _lib.VPConfig, 'mansy_gemm_epilogue': _lib.GemmEpilogue, 'mansy_env_tables': _lib.EnvTables,
             'mansy_env_episode_log': _lib.EpisodeLog, 'mansy_attn_shape': _lib.AttnShape, 'mansy_xg_handle': _lib.XgHandle,
             'mansy_comm_id': _lib.CommId}
    hdr = os.path.join(ROOT, 'include', 'mansy_hip.h')
    text = open(hdr).read()
    assert set(re.findall(r'typedef struct (mansy_\w+)', text)) == set(pairs)      # no struct of the header is left unmirrored
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{hdr}"', 'int main(void) {']
    for cname, cls in pairs.items():
        lines.append(f'  printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, _ in cls._fields_:
            lines.append(f'  printf("{cname}.{fname} %zu\\n", offsetof({cname}, {fname}));')
    lines += ['  return 0;', '}']
    src, exe = tmp_path / 'layout.c', tmp_path / 'layout'
    src.write_text('\n'.join(lines))
    subprocess.run(['gcc', '-std=c11', '-o', str(exe), str(src)], check=True, capture_output=True)   # an unknown field name fails here
    got = dict(l.split() for l in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for cname, cls in pairs.items():
        assert int(got[cname]) == ctypes.sizeof(cls), (cname, got[cname], ctypes.sizeof(cls))
        for fname, _ in cls._fields_:
            assert int(got[f'{cname}.{fname}']) == getattr(cls, fname).offset, (cname, fname)
        # and the header has no field the mirror lacks: count the declarators of the struct body
        body = re.search(r'typedef struct %s \{(.*?)\} %s;' % (cname, cname), text, re.S).group(1)
        body = re.sub(r'/\*.*?\*/', '', body, flags=re.S)
        n_fields = sum(len(decl.split(',')) for decl in body.split(';') if decl.strip())
        assert n_fields == len(cls._fields_), (cname, n_fields, len(cls._fields_))


def test_ctypes_prototypes_match_header_signatures():
    """Argument count and scalar-vs-pointer class of every prototype in _lib._PROTOS against the declaration in the header
    (a short or mistyped argument list is undefined behaviour that no compute test would necessarily catch)."""
    from mansy_immersivevideostreaming_amd import _lib
    src = open(os.path.join(ROOT, 'include', 'mansy_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    decls = {m.group(2): (m.group(1).strip(), m.group(3)) for m in re.finditer(r'^([A-Za-z_][\w \*]*?)\b(mansy_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;', src, re.M | re.S)}
    assert set(decls) == set(_lib._PROTOS), set(decls) ^ set(_lib._PROTOS)
    for name, (ret, args) in decls.items():
        args = [a.strip() for a in args.replace('\n', ' ').split(',')] if args.strip() not in ('', 'void') else []
        proto = _lib._PROTOS[name]
        assert len(args) == len(proto), (name, len(args), len(proto))
        for a, t in zip(args, proto):
            is_ptr_c = '*' in a or '[' in a or 'mansy_bn_sync_fn' in a
            is_ptr_py = t in (ctypes.c_void_p, ctypes.c_char_p) or hasattr(t, '_type_') and isinstance(t._type_, type) or 'CFunctionType' in str(t.__mro__)
            if is_ptr_c:
                assert is_ptr_py, (name, a, t)
            else:
                assert t in (ctypes.c_int, ctypes.c_uint, ctypes.c_uint32, ctypes.c_float, ctypes.c_double, ctypes.c_longlong, ctypes.c_ulonglong,
                             ctypes.c_size_t, ctypes.c_int64, ctypes.c_uint64), (name, a, t)
                want = {'float': ctypes.c_float, 'double': ctypes.c_double}.get(a.split()[0] if a.split()[0] != 'const' else a.split()[1])
                if want is not None:
                    assert t is want, (name, a, t)
                if a.split()[0] in ('int', 'uint32_t', 'unsigned'):
                    assert ctypes.sizeof(t) == 4, (name, a, t)
                if 'long long' in a or 'int64_t' in a or 'size_t' in a:
                    assert ctypes.sizeof(t) == 8, (name, a, t)
