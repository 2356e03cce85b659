#!/usr/bin/env python3
"""bench.py -- VP-Transformer training throughput (BASELINE.json configs[1]) on N MI355X GPUs.

A "step" = one pass of run_models.py:37-44 over one synthetic batch of B=4096 trajectories per GPU
(MTIO mix, zero-grad, forward, MTIO loss, backward, AdamW), executed by libmansy_hip.so.
Inputs are resident in HBM before the timed region.  N>1: one process per GPU (torch.distributed /
RCCL), data parallel over trajectories ("weak" scaling: 4096 per GPU), ONE all-reduce of the flat
gradient buffer per step.  Both launch forms work: under `python -m torch.distributed.run --nproc-per-node N ... bench.py
--gpus N` the ranks come from the environment; a plain `python bench.py --gpus N` starts its own N rank processes
(spawn_ranks: fresh children, before this process has touched the GPU).  The line carries `dist` = what the process group
itself reported (backend, world size, an all-reduce head count); a run whose world size is not --gpus refuses to print.

Prints ONE JSON line (rank 0).  Extra objects:
  roofline     dominant kernel = gemm_f32_dma_kernel (exact-fp32 MFMA, LDS-DMA staged).  achieved = exact GEMM FLOPs
               (sum 2MNK over launches; KV-cached decoder => this IS the algorithmic minimum of SURVEY
               8d up to the attention/elementwise terms) / summed launch duration from HIP event pairs attached to
               each dispatch on the launch stream (the kernel's own begin / end) in a separate instrumented leg of the same workload.
               `model_frac` = trajectories/s x 0.522 GFLOP / peak (whole-step MFMA utilisation, north_star).
  cpu_baseline the oracle (CPU restatement, torch fp32, autograd) timed on the host cores on a bounded
               sample (B=32 steps, thread counts 1/8/16/32/all swept inside a ~15 s budget, best reported), kind "port".
  precision_modes  the same step with the dense products in the split-bf16 modes (bf16x3, bf16x6: csrc/gemm_bf16s.hip), each
               with its own roofline object priced against the dense BF16 MFMA peak.  The fp32 line stays the headline.
  secondary    PPO env-steps/s (configs[2]) with its own roofline object (MFMA fraction on 11.5 MFLOP, HBM fraction on 29 KB per
               env-step, the dominant kernel's live launch timing, and the launch-latency floor of the cycle).
"""
import argparse
import ctypes
import json
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_TRAJ = 0.522e9          # fwd+bwd algorithmic FLOPs per trajectory (d=ff=512, S=T=10, 2+2 layers)
PEAK_F32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: FP32 matrix peak (spec)
PEAK_BF16_MFMA_TFLOPS = 2500.0   # MI355X_MICROARCH.md: dense BF16 MFMA peak (spec; random-data loops hold ~1.25-1.5 PF: DVFS)
PEAK_HBM_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E
PPO_FLOP_PER_ENV_STEP = 11.5e6   # rollout 0.523 + update 10.95 MFLOP per env-step, as written
PPO_BYTES_PER_ENV_STEP = 29e3    # 3 116 B observation written + ~8.2 reads of it across the update passes
LAUNCH_FLOOR_US = 5.0            # dependent-launch floor on this chip (K -> 0 intercept of a [4096,512] product)


def replica_spread(flat, world):
    """Data-parallel sanity figure (after the timed region): max over ranks minus min over ranks of a checksum of the
    parameters.  Identical replicas (same initial weights, averaged gradients, synchronised BatchNorm statistics) give 0."""
    import torch
    import torch.distributed as dist
    if world <= 1:
        return 0.0
    s = flat.double().sum().reshape(1)
    a = flat.double().abs().sum().reshape(1)
    hi, lo = torch.cat([s, a]), torch.cat([s, a])
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    return float((hi - lo).abs().max().item())


def synthetic_trajectories(B, S, T, seed=5):
    """SURVEY 8(d) C2 inputs: smooth random walks on the unit torus, S + 1 + T samples (history, current, future), fp32."""
    import torch
    g = torch.Generator().manual_seed(seed)
    p0 = torch.rand(B, 1, 2, generator=g)
    steps = torch.randn(B, S + T, 2, generator=g) * 0.02
    traj = torch.cat([p0, p0 + torch.cumsum(steps, dim=1)], dim=1)
    traj = traj - torch.floor(traj)
    return traj[:, :S].contiguous(), traj[:, S:S + 1].contiguous(), traj[:, S + 1:].contiguous()


def cpu_baseline(seconds=15.0):
    """The VP oracle (KV-cached restatement, torch fp32 autograd + AdamW) on the host cores: thread counts are swept inside the
    budget and the best is reported (320-token GEMMs do not scale to every core of a 2-socket host; the round-1 line ran all
    128 threads and understated the CPU)."""
    import torch
    from oracle import vp_oracle as vo
    B, S, T, d = 32, 10, 10, 512
    sd = vo.make_state_dict(d, 5, bias=True)
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()
              if v.dtype.is_floating_point and 'running_' not in k and k != 'positional_embedding.pe'}
    full = dict(sd)
    full.update(params)
    orc = vo.VPOracle(full, fut_window=T)
    h, c, f = vo.synthetic_trajectories(B, S, T, seed=5)
    m = {k: torch.zeros_like(v) for k, v in params.items()}
    v2 = {k: torch.zeros_like(v) for k, v in params.items()}
    step_no = [0]

    def one_step():
        src, cur, gt = vo.mtio_mix(h, c, f, 3, True, None)
        loss = orc.loss_function(orc.process_src_current(src, cur, train=True), gt)
        for p in params.values():
            p.grad = None
        loss.backward()
        step_no[0] += 1
        with torch.no_grad():
            for k, p in params.items():
                p1, m[k], v2[k] = vo.adamw_step(p, p.grad, m[k], v2[k], step=step_no[0])
                p.copy_(p1)

    all_threads = torch.get_num_threads()
    counts = sorted({n for n in (1, 8, 16, 32, all_threads) if n <= all_threads})
    one_step()                                            # warm-up (allocator, autograd graph caches)
    sweep, best = {}, (0.0, all_threads, 0, 0.0)
    per = seconds / len(counts)
    for nt in counts:
        torch.set_num_threads(nt)
        n, t0 = 0, time.time()
        while True:
            one_step()
            n += 1
            if time.time() - t0 > per and n >= 2:
                break
        dt = time.time() - t0
        sweep[str(nt)] = round(n * B / dt, 2)
        if n * B / dt > best[0]:
            best = (n * B / dt, nt, n, dt)
    torch.set_num_threads(all_threads)
    return {'value': round(best[0], 2), 'unit': 'trajectories/s', 'cores': best[1], 'kind': 'port',
            'sample': f'{best[2]} train steps of B={B} (S=T=10, d=512, 2+2 layers, dropout off, KV-cached oracle) in {best[3]:.1f}s on '
                      f'{best[1]} threads (best of the sweep)',
            'thread_sweep_traj_per_s': sweep, 'host_threads': all_threads,
            # the reference re-runs the decoder on the growing target (55 token-positions instead of 10) and re-projects the cross
            # K/V at every step: 1.945 vs 0.522 GFLOP per trajectory fwd+bwd -- the port above does the minimal work
            'reference_as_written_flop_factor': round(1.945 / 0.522, 2),
            'survey_container_reference_traj_per_s': {'value': 84, 'cores': 8, 'note': 'imported reference, B=32, SURVEY section 6'}}


def _pmc_traffic(pattern='r*_pmc_gemm.json'):
    """HBM-side bytes per GEMM launch from the newest committed rocprofv3 PMC aggregate (profiles/<round tag>_pmc_gemm*.json,
    written by tools/pmc_aggregate.py from separate FETCH_SIZE / WRITE_SIZE passes, FETCH_SIZE x2 per the gfx950 correction);
    PMC cannot be collected from inside this process.  Returns (bytes per launch, file name, stale): the file is named in the
    bench line, and `stale` says whether the GEMM sources it was collected on (their digest is stamped into the file) are the
    ones running now -- counters of an older kernel must not pass for this build's."""
    import glob
    from mansy_immersivevideostreaming_amd import build_ext
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', pattern)))
    try:
        rec = json.load(open(files[-1]))
        stale = rec.get('gemm_source_digest') != build_ext.gemm_source_digest()
        return rec['traffic_bytes_per_launch'], 'profiles/' + os.path.basename(files[-1]), stale
    except Exception:
        return None, None, None


def _ppo_torch_kernels_per_cycle():
    """Kernels per PPO cycle that are NOT this library's (torch's own: the uniforms of the Categorical sampling, slab copies, index
    arithmetic of the host mirror), from the newest committed rocprofv3 kernel-stats file of the cycle next to its meta file
    (profiles/r*_ppo_kernel_stats.csv + .meta.json = tools/gpu_prof_ppo.sh: the cycle count of that profile and the digest of the
    sources it ran).  The library's own launches are counted LIVE (mansy_prof_launch_count); this is the only profile-derived part of
    the launch count and it carries a stale flag."""
    import csv
    import glob
    from mansy_immersivevideostreaming_amd import build_ext
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_ppo_kernel_stats.csv')))
    try:
        meta = json.load(open(files[-1][:-4] + '.meta.json'))
        rows = list(csv.DictReader(open(files[-1])))
        theirs = ('at::', 'rocclr', 'rocprim', 'hipcub')            # torch's kernels + the HIP runtime's blit kernels (copy / fill)
        other = sum(int(r['Calls']) for r in rows if any(t in r['Name'] for t in theirs))
        return round(other / float(meta['cycles']), 1), 'profiles/' + os.path.basename(files[-1]), meta.get('source_digest') != build_ext.source_digest()
    except Exception:
        return None, None, None


def _ppo_pmc_traffic():
    """HBM-side bytes per env-step of the whole PPO cycle from the newest committed PMC aggregate (profiles/r*_ppo_pmc.json, written by
    tools/gpu_prof_ppo.sh from separate FETCH_SIZE / WRITE_SIZE passes over every kernel of the cycle, FETCH_SIZE x2 per the gfx950
    correction) -> (bytes per env-step, file, stale): stale = the library sources have changed since the counters were taken."""
    import glob
    from mansy_immersivevideostreaming_amd import build_ext
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_ppo_pmc.json')))
    try:
        rec = json.load(open(files[-1]))
        return rec['traffic_bytes_per_env_step'], 'profiles/' + os.path.basename(files[-1]), rec.get('source_digest') != build_ext.source_digest()
    except Exception:
        return None, None, None


def _gemm_prof(L, fn, reps):
    """A HIP event pair attached to every GEMM dispatch (hipExtLaunchKernelGGL: the kernel's own begin / end on rank 0's launch stream) of
    `reps` calls of fn: (ms, launches, exact FLOPs)."""
    from mansy_immersivevideostreaming_amd._lib import check
    import torch
    check(L.mansy_prof_gemm_enable(1), 'prof_enable')
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    ms, n, fl = ctypes.c_double(), ctypes.c_longlong(), ctypes.c_double()
    check(L.mansy_prof_gemm_collect(ctypes.byref(ms), ctypes.byref(n), ctypes.byref(fl)), 'prof_collect')
    check(L.mansy_prof_gemm_enable(0), 'prof_disable')
    return ms.value, n.value, fl.value


QOE_TRAIN = ((7, 1, 1), (1, 7, 1), (1, 1, 7), (3, 3, 3))          # bitrate_selection/config.yml:142 (train / valid preferences)
QOE_TEST = ((5, 1, 3), (2, 4, 3), (1, 3, 5), (4, 4, 1))           # config.yml:144 (test preferences)


def _ppo_policy(dev, rank=0, precision=None):
    """run_mansy.py:205-251's nets + PPOPolicy with the reference's hyper-parameters, random-init (orthogonal, as the reference)."""
    import numpy as np
    import torch
    from mansy_immersivevideostreaming_amd.bitrate_selection.models import mansy as mm
    from mansy_immersivevideostreaming_amd.bitrate_selection.models.mansy_ppo import PPOPolicy

    class A:
        use_identifier, lamb = True, 0.5
    torch.manual_seed(5)
    np.random.seed(5 + rank)
    fn = mm.FeatureNet(8, 64, 5, 128, device=dev)
    actor, critic = mm.Actor(fn, 1280, 128, 15, dev), mm.Critic(fn, 1280, 128, dev)
    ident = mm.QoEIdentifier(mm.QoEIdentifierFeatureNet(8, 64, 5, 15, 128, device=dev), 1280, 128, dev)
    mm.orthogonal_init(actor, critic)
    mm.orthogonal_init(ident)
    optim = torch.optim.Adam(actor.parameters(), lr=5e-4, weight_decay=1e-2)
    ioptim = torch.optim.Adam(ident.parameters(), lr=1e-4, weight_decay=1e-2)
    pol = PPOPolicy(actor, critic, optim, None, discount_factor=0.95, max_grad_norm=1.0, eps_clip=0.2, vf_coef=0.5, ent_coef=0.02,
                    reward_normalization=1, advantage_normalization=1, value_clip=1, gae_lambda=0.95, action_space=15, args=A(),
                    identifier=ident, identifier_optim=ioptim).to(dev)
    if precision is not None:
        pol.engine.precision = precision
    return pol


PPO_BLOCKS = 5      # a PPO leg = the MEDIAN of this many timed blocks of `cycles` cycles (6 cycles are 11 ms: one 1-ms hiccup of the host thread moved a single-block figure by 10 %)


def _timed_blocks(cycle, cycles, blocks=PPO_BLOCKS, barrier=None):
    """`blocks` back-to-back timed blocks of `cycles` calls of cycle(), each bracketed by a device sync (and `barrier()`), Python's collector off.  Returns
    ([(wall s, host-enqueue s)] per block, last result).  The host never gets a whole block ahead of the device, so the ring of pinned staging buffers of the
    graph-replayed update never makes it wait."""
    import gc
    import torch
    gc.collect()
    out, res = [], None
    gc.disable()
    try:
        for _ in range(blocks):
            if barrier:
                barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(cycles):
                res = cycle()
            t_host = time.perf_counter() - t0      # the host has enqueued everything (it runs ahead of the GPU unless the cycle is host-bound)
            torch.cuda.synchronize()
            if barrier:
                barrier()
            out.append((time.perf_counter() - t0, t_host))
    finally:
        gc.enable()
    return out, res


def _median_block(blocks):
    """(median wall time, median host-enqueue time) over the blocks -- each on its own: a block's host time moves independently of its wall time."""
    return sorted(b[0] for b in blocks)[len(blocks) // 2], sorted(b[1] for b in blocks)[len(blocks) // 2]


def _ppo_cycle_time(pol, dev, cycles, warmup, n_env=256, steps_per_env=16, qoe_weights=QOE_TRAIN, seed=5, tables=None):
    """(ms per cycle, library launches per cycle incl. the rollout graph's) of collect -> train_identifier (2 rounds) -> relabel + PPO update."""
    import torch
    from mansy_immersivevideostreaming_amd._lib import lib
    from mansy_immersivevideostreaming_amd.bitrate_selection.envs.mansy_env import EnvTables, MANSYVecEnv
    from mansy_immersivevideostreaming_amd.bitrate_selection.models.mansy_ppo import RolloutBuffer, VecCollector
    if tables is None:
        tables = EnvTables.synthetic(dev, seed=5, qoe_weights=qoe_weights, train_identifier_reward=True, n_sample=max(240, n_env))
    venv = MANSYVecEnv(tables, n_env, seed=seed, index_offset=0, worker_num=n_env)
    col = VecCollector(pol, venv, seed=seed)
    buf = RolloutBuffer(steps_per_env, n_env, dev)

    def cycle():
        col.collect(steps_per_env * n_env, buf)
        pol.train_identifier(buf, 2, verbose=False)
        return pol.update(0, buf, is_train=True, batch_size=512, repeat=2)
    for _ in range(warmup):
        cycle()
    # Python's cyclic collector must not run inside the timed cycles: the previous leg's policy (an nn.Module: reference cycles) holds captured hipGraphs and
    # their memory pools, and tearing those down costs tens of ms whenever the collector happens to fire (seen as one 25-70 ms cycle in a leg of six)
    n0 = lib().mansy_prof_launch_count() + pol.graph_launches
    blocks, res = _timed_blocks(cycle, cycles)
    n1 = lib().mansy_prof_launch_count() + pol.graph_launches      # direct launches + the ones the update half's graph replays re-ran
    dt, t_host = _median_block(blocks)
    n1 = n0 + (n1 - n0) / float(len(blocks))
    import numpy as np
    loss = float(np.mean(res['loss']))
    _ppo_cycle_time.host_ms = t_host / cycles * 1e3
    _ppo_cycle_time.graph_replays = pol.graph_replays
    return dt / cycles * 1e3, (n1 - n0) / float(cycles) + (col.graph_launches if col.use_graph and col._graph is not None else 0), loss


def bench_ppo_dp_form(dev, mdist, cycles=6, warmup=2, vp_leg=None):
    """VERDICT r04 #1a: what data parallelism costs a rank BEFORE any wire time, measured on one MI355X.  The same PPO cycle three ways:
    fused (single process: gradient norm, clip and Adam ride on the step's own launches), and the data-parallel FORM at world 1 with the
    average really issued -- the hand-written peer kernel on a one-rank context (csrc/xgmi.hip) and the library collective (RCCL AVG over a
    one-rank process group).  `implied_ceiling_8gpu` = 8 x fused / dp: the best 1 -> 8 scaling this form allows if the wire were free."""
    import torch
    import torch.distributed as dist
    out = {'workload': '256 envs x 16 steps, identifier 2 rounds + relabel + PPO update (minibatch 512, repeat 2): 16 + 2 gradient averages per cycle',
           'cycles': cycles, 'timing': 'median of %d blocks of %d cycles' % (PPO_BLOCKS, cycles)}
    ms, nl, _ = _ppo_cycle_time(_ppo_policy(dev), dev, cycles, warmup)
    out['fused'] = {'ms_per_cycle': round(ms, 3), 'library_launches_per_cycle': round(nl, 1), 'host_enqueue_ms_per_cycle': round(_ppo_cycle_time.host_ms, 3),
                    'update_graph_replays': _ppo_cycle_time.graph_replays}
    for key, in_slot in (('dp_peer_kernel', True), ('dp_peer_kernel_copy_form_r04', False)):      # slot form: one library call per step, as the fused step
        pol = _ppo_policy(dev)
        pol.peer_in_slot = in_slot      # True (round 5): gradients produced straight in the exchange slot; False: copied into it by the collective launch
        pol.set_data_parallel(1, None, peer=True, force=True)
        ms_x, nl_x, _ = _ppo_cycle_time(pol, dev, cycles, warmup)
        pol._check_peers()
        out[key] = {'ms_per_cycle': round(ms_x, 3), 'library_launches_per_cycle': round(nl_x, 1), 'host_enqueue_ms_per_cycle': round(_ppo_cycle_time.host_ms, 3),
                    'update_graph_replays': _ppo_cycle_time.graph_replays, 'vs_fused': round(ms_x / ms, 3),
                    'us_per_average': round((ms_x - ms) * 1e3 / 18, 2), 'implied_ceiling_8gpu': round(8 * ms / ms_x, 2)}
    try:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if 'MASTER_PORT' not in os.environ:
            import socket
            with socket.socket() as s:
                s.bind(('127.0.0.1', 0))
                os.environ['MASTER_PORT'] = str(s.getsockname()[1])
        os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1'); os.environ.setdefault('LOCAL_RANK', '0')
        mdist.init_process_group(backend='nccl', force=True)
        pol = _ppo_policy(dev)
        pol.set_data_parallel(1, mdist.make_grad_sync(1, force=True), peer=False, force=True)
        ms_r, nl_r, _ = _ppo_cycle_time(pol, dev, cycles, warmup)
        out['dp_rccl'] = {'ms_per_cycle': round(ms_r, 3), 'library_launches_per_cycle': round(nl_r, 1), 'host_enqueue_ms_per_cycle': round(_ppo_cycle_time.host_ms, 3), 'vs_fused': round(ms_r / ms, 3),
                          'us_per_average': round((ms_r - ms) * 1e3 / 18, 2), 'implied_ceiling_8gpu': round(8 * ms / ms_r, 2),
                          'note': 'launch counts exclude RCCL\'s own kernels'}
        # the same library collective through the library's OWN communicator wrapper (mansy_comm_* / mansy_allreduce_*): the step is one call again
        comm = mdist.RcclComm(1, 0, dev)
        pol = _ppo_policy(dev)
        pol.set_data_parallel(1, None, peer=False, force=True, comm=comm)
        pol.graph_update = True          # the update half incl. ncclAllReduce captured into the graph (forced: 'auto' keeps library collectives out of graphs)
        ms_c, nl_c, _ = _ppo_cycle_time(pol, dev, cycles, warmup)
        out['dp_rccl_one_call'] = {'ms_per_cycle': round(ms_c, 3), 'library_launches_per_cycle': round(nl_c, 1), 'host_enqueue_ms_per_cycle': round(_ppo_cycle_time.host_ms, 3),
                                   'vs_fused': round(ms_c / ms, 3), 'us_per_average': round((ms_c - ms) * 1e3 / 18, 2), 'implied_ceiling_8gpu': round(8 * ms / ms_c, 2),
                                   'update_graph_replays': pol.graph_replays,
                                   'note': 'mansy_ppo_minibatch_step(..., sync = mansy_comm context): ncclAllReduce(avg) + norm launch inside the step call; update half captured as hipGraphs with the collective inside (graph_update = True)'}
        comm.close()
        if vp_leg is not None:          # the VP step in ITS data-parallel form over the same one-rank RCCL group
            out['vp_step'] = vp_leg()
        dist.destroy_process_group()
    except Exception as e:          # noqa: BLE001 -- a box without a usable RCCL: say so instead of losing the line
        out['dp_rccl'] = {'error': str(e)[:200]}
    return out


def vp_dp_form_leg(model, opt, h, c, f, dev, mdist, steps=8, warmup=3):
    """The VP train step (B = 4096) in its data-parallel form at world 1 over a one-rank RCCL group, everything issued: the two SyncBN statistics
    all-reduces from inside the engine's hook, the flat-gradient AVG with its decoder-side two thirds on the side stream under the encoder backward
    (dist.OverlappedGradSync), AdamW as a launch of its own behind the collective.  The one real rank stands for two identical ones (the hook doubles
    the reduced sums: tools/rccl_selftest.py), so the arithmetic equals the single-process step."""
    import torch
    import torch.distributed as dist

    def timed(step):
        for _ in range(warmup):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3
    ms_plain = timed(lambda: model.train_step(h, c, f, opt))

    def bn_allreduce(t):
        dist.all_reduce(t)
        t.mul_(2.0)
    model.set_data_parallel(2, allreduce=bn_allreduce)
    osync = mdist.OverlappedGradSync(1, dev, force=True)
    try:
        ms_dp = timed(lambda: model.train_step(h, c, f, opt, grad_sync=osync))
    finally:
        model.set_data_parallel(1)
    return {'workload': 'VP train step B=4096 fp32: SyncBN hook (2 x 1024 doubles) + 36.8 MB flat-gradient AVG (tail overlapped) + AdamW behind the collective, one-rank RCCL group',
            'ms_per_step_single_process': round(ms_plain, 3), 'ms_per_step_dp_form': round(ms_dp, 3), 'vs_single_process': round(ms_dp / ms_plain, 4)}


def bench_ppo_c5(dev, cycles=5, warmup=2):
    """BASELINE configs[4] on one GPU (its per-GPU share): the PPO cycle over the 8-preference table (config.yml:141-144: 4 train + 4 test
    vectors in one vectorised environment), identifier training on, dense products in the bf16 modes ("bf16 MFMA")."""
    out = []
    for mode in ('f32', 'bf16x3', 'bf16'):
        ms, nl, loss = _ppo_cycle_time(_ppo_policy(dev, precision=mode), dev, cycles, warmup, qoe_weights=QOE_TRAIN + QOE_TEST)
        eps = 256 * 16 / ms * 1e3
        nprod = {'f32': 1, 'bf16x3': 3, 'bf16': 1}[mode]
        peak = PEAK_F32_MFMA_TFLOPS if mode == 'f32' else PEAK_BF16_MFMA_TFLOPS
        out.append({'dtype': mode, 'metric': 'PPO env-steps/sec', 'value': round(eps, 1), 'unit': 'env-steps/s', 'ms_per_cycle': round(ms, 3),
                    'library_launches_per_cycle': round(nl, 1), 'final_loss': loss,
                    'roofline': {'bound': 'mfma', 'achieved': round(eps * PPO_FLOP_PER_ENV_STEP * nprod / 1e12, 3), 'peak': peak, 'unit': 'TFLOP/s',
                                 'frac': round(eps * PPO_FLOP_PER_ENV_STEP * nprod / 1e12 / peak, 5),
                                 'launch_floor_ms': round(nl * LAUNCH_FLOOR_US * 1e-3, 3),
                                 'note': 'the cycle is a chain of ~200 dependent launches of 5-12 us: precision changes the arithmetic of its products, not its time'}})
    return {'config': {'workload': 'BASELINE configs[4], per-GPU share: 256 envs over the 8-preference table (4 train + 4 test QoE vectors), identifier '
                                   'training (2 rounds) + relabel + PPO update, 16 steps per env per collect, synthetic Jin2022/4G-shaped tables'},
            'modes': out}


def bench_vp_small(dev, steps=30, warmup=5):
    """The reference's REAL batch sizes: BASELINE configs[0] on the GPU (B = 32, hist 10, pred 10) and the README's training
    command (B = 512, hist 5, pred 15: README.md:139, run_models.py:143,196).  Launch-bound: ms per step, library launches per step."""
    import numpy as np
    import torch
    from mansy_immersivevideostreaming_amd._lib import lib
    from mansy_immersivevideostreaming_amd.viewport_prediction.models import ViewportTransformerMTIO, FusedAdamW
    out = []
    for name, B, S, T in (('configs[0] on the GPU', 32, 10, 10), ('README training command', 512, 5, 15)):
        torch.manual_seed(5); random.seed(5); np.random.seed(5)
        m = ViewportTransformerMTIO(in_channel=2, fut_window=T, d_model=512, dim_feedforward=512, device=dev).to(dev)
        m.train()
        opt = FusedAdamW(m, lr=1e-4)
        h, c, f = (t.to(dev) for t in synthetic_trajectories(B, S, T, seed=5))
        for _ in range(warmup):
            m.train_step(h, c, f, opt)
        torch.cuda.synchronize()
        # three blocks of `steps` steps, the MEDIAN block reported (an auxiliary leg of ~0.1 s: one host or box hiccup -- a 4.5 ms B = 32 "step" has been seen
        # once in ~20 runs of this leg at 2.75 -- would otherwise be the number)
        blocks = []
        n0 = lib().mansy_prof_launch_count()
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(steps):
                loss = m.train_step(h, c, f, opt)
            torch.cuda.synchronize()
            blocks.append((time.perf_counter() - t0) / steps * 1e3)
        n1 = lib().mansy_prof_launch_count()
        ms = sorted(blocks)[1]
        m.eval()
        with torch.no_grad():
            for _ in range(3):
                m.sample(h, c)
            torch.cuda.synchronize()
            s0 = lib().mansy_prof_launch_count()
            t1 = time.perf_counter()
            for _ in range(steps):
                m.sample(h, c)
            s1 = lib().mansy_prof_launch_count()
            torch.cuda.synchronize()
            ms_s = (time.perf_counter() - t1) / steps * 1e3
        out.append({'name': name, 'B': B, 'S': S, 'T': T, 'ms_per_step': round(ms, 3), 'trajectories_per_s': round(B / ms * 1e3, 1),
                    'ms_per_step_blocks': [round(b, 3) for b in blocks], 'timing': f'median of 3 blocks of {steps} steps',
                    'library_launches_per_step': round((n1 - n0) / (3 * steps), 1), 'final_loss': float(loss.item()),
                    'sample_ms': round(ms_s, 3), 'sample_trajectories_per_s': round(B / ms_s * 1e3, 1),
                    'sample_launches': round((s1 - s0) / steps, 1),
                    'model_frac_of_f32_peak': round(B / ms * 1e3 * (FLOP_PER_TRAJ if (S, T) == (10, 10) else float('nan')) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4)
                    if (S, T) == (10, 10) else None})
    return out


REAL_TABLES = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tests', 'golden', 'env_tables_jin2022_4g.npz')


def _ppo_tables(dev, kind, n_env_total):
    """'real': the reference's Jin2022 x 4G TRAIN split (18 videos x 45 users x 24 traces x 4 preferences = the 72-episode catalogue of
    generate_environment_samples, utils/common.py:60-84), packed by tools/gen_golden_tables_full.py out of the reference's own loaders --
    BASELINE configs[2] "Jin2022 tiles x 4G bandwidth traces".  'synthetic': same-shape random tables."""
    from mansy_immersivevideostreaming_amd.bitrate_selection.envs.mansy_env import EnvTables
    if kind == 'real':
        return EnvTables.from_file(REAL_TABLES, 'train', dev, use_identifier=True)
    return EnvTables.synthetic(dev, seed=5, train_identifier_reward=True, n_sample=max(240, n_env_total))


def bench_ppo(rank, world, dev, mdist, cycles=5, warmup=2, n_env=256, steps_per_env=16, rollout_probe=True, tables_kind='real'):
    """PPO env-steps/s (BASELINE configs[2]/[3]): 256 vectorised trace-sim envs per GPU on the reference's real Jin2022 x 4G train tables
    (`tables_kind='synthetic'`: bench-shaped random tables), one cycle = collect 16 steps/env (4096 transitions/GPU) -> train_identifier
    (2 rounds) -> relabel -> PPO update (minibatch 512, repeat 2), i.e. run_mansy.py --train --train-identifier --use-identifier with
    step_per_collect=4096."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from mansy_immersivevideostreaming_amd.bitrate_selection.envs.mansy_env import EnvTables, MANSYVecEnv
    from mansy_immersivevideostreaming_amd.bitrate_selection.models.mansy_ppo import RolloutBuffer, VecCollector
    pol = _ppo_policy(dev, rank)
    # The 16 + 2 gradient averages of a cycle are latency-bound (1.7 MB / 1.05 MB, each on the critical path).  Two implementations: the
    # library collective (RCCL) and the hand-written one-shot all-reduce over hipIpc-mapped peer memory (csrc/xgmi.hip).  Default: build
    # the second, check it against the first on the same data, time both on THIS machine's links and keep the faster correct one
    # (dist.probe_peer_grad_sync; any set-up failure, wrong result or timed-out wait on any rank -> the library collective on every rank).
    # MANSY_PEER_SYNC=1 / 0 forces one or the other.
    # 'slot': forced peer kernel in the exchange-slot form (the form 'auto' takes once its probe has agreed: the functional 8-rank test uses it);
    # '1': forced peer kernel, un-probed, hence the copy form (PPOPolicy.set_data_parallel).
    mode = os.environ.get('MANSY_PEER_SYNC', 'auto')
    peer = False if world <= 1 or mode == '0' else (True if mode in ('1', 'slot') else 'auto')
    pol.set_data_parallel(world, mdist.make_grad_sync(world), peer=peer, in_slot=True if mode == 'slot' else None)
    rep = pol.grad_sync_report
    rccl_direct = None
    if world > 1 and rep.get('chosen') != 'peer' and dist.get_backend() == 'nccl' and os.environ.get('MANSY_RCCL_DIRECT', '1') != '0':
        # the library collective it is: through the library's own communicator (mansy_comm_*), so that the data-parallel step stays ONE call per step
        # (1.07x the single-process cycle at world 1 against 1.2x through three calls); any rank that cannot -> torch.distributed on every rank
        comm, why = mdist.RcclComm.try_create(world, rank, dev)
        if comm is not None:
            pol.set_data_parallel(world, mdist.make_grad_sync(world), peer=False, comm=comm)
            pol.grad_sync_report = rep
            rccl_direct = 'mansy_comm (one call per step)'
        else:
            rccl_direct = f'torch.distributed (mansy_comm not taken: {why})'
    if world <= 1:
        sync_desc = 'none'
    elif rep.get('chosen') == 'peer':
        sync_desc = 'peer-memory one-shot (csrc/xgmi.hip)' + ('' if peer is True else
                                                             f"; probe: peer {rep.get('us_peer')} us vs library {rep.get('us_library')} us per {rep.get('floats')}-float average")
    else:
        sync_desc = ('RCCL all-reduce via ' + rccl_direct if rccl_direct else 'torch.distributed all_reduce') + ('' if peer is False else f"; probe: {rep.get('reason')}" + (
            f" (peer {rep.get('us_peer')} us vs library {rep.get('us_library')} us)" if 'us_peer' in rep else ''))
    if tables_kind == 'real' and not os.path.exists(REAL_TABLES):
        tables_kind = 'synthetic'
    tables = _ppo_tables(dev, tables_kind, n_env * world)
    off, wnum = mdist.shard_envs(n_env, rank, world)
    shards = [[off, wnum]]
    if world > 1:                               # every rank's (first global environment, global worker count): the line shows the partition
        shards = [None] * world
        dist.all_gather_object(shards, [off, wnum])
    venv = MANSYVecEnv(tables, n_env, seed=5, index_offset=off, worker_num=wnum)
    col = VecCollector(pol, venv, seed=5 + rank)
    buf = RolloutBuffer(steps_per_env, n_env, dev)

    def cycle():
        col.collect(steps_per_env * n_env, buf)
        pol.train_identifier(buf, 2, verbose=False)
        return pol.update(0, buf, is_train=True, batch_size=512, repeat=2)
    for _ in range(warmup):
        cycle()
    from mansy_immersivevideostreaming_amd._lib import lib
    launches0 = lib().mansy_prof_launch_count() + pol.graph_launches
    blocks, res = _timed_blocks(cycle, cycles, barrier=dist.barrier if world > 1 else None)      # (the update half replays captured graphs from the third cycle on)
    launches1 = lib().mansy_prof_launch_count() + pol.graph_launches
    launches1 = launches0 + (launches1 - launches0) / float(len(blocks))
    if world > 1:          # a block's time = the slowest rank's; then the median block
        tb = torch.tensor([b[0] for b in blocks], device=dev, dtype=torch.float64)
        dist.all_reduce(tb, op=dist.ReduceOp.MAX)
        dt_max = sorted(float(x) for x in tb)[len(blocks) // 2]
        dt, t_host = _median_block(blocks)
    else:
        dt, t_host = _median_block(blocks)
    per_rank = None
    if world > 1:
        # every rank's own view: cycle time to its last sync and the time its host needed to ENQUEUE the cycles (a straggling enqueue thread
        # is what every other rank's peer kernel then waits for)
        mine = torch.tensor([dt / cycles * 1e3, t_host / cycles * 1e3], device=dev, dtype=torch.float64)
        allv = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allv, mine)
        per_rank = {'ms_per_cycle': [round(float(v[0]), 3) for v in allv], 'host_enqueue_ms_per_cycle': [round(float(v[1]), 3) for v in allv]}
        dt = dt_max
    # the cycle in its data-parallel FORM on a ONE-rank peer context, on every rank at once (no wire, no waiting for anyone): what is left of the
    # N-rank cycle after subtracting it is the wire + skew term w of DESIGN section 6, per average
    wire = None
    if world > 1:
        ms1, err1 = -1.0, ''
        try:                            # rank-local work only inside the try: every rank reaches the collectives below whatever happened here
            pol1 = _ppo_policy(dev, rank)
            pol1.set_data_parallel(1, None, peer=True, force=True, in_slot=True)
            ms1, _, _ = _ppo_cycle_time(pol1, dev, cycles, 3, n_env=n_env, steps_per_env=steps_per_env, tables=tables)
            pol1._check_peers()
        except Exception as e:          # noqa: BLE001
            ms1, err1 = -1.0, str(e)[:200]
        t1 = torch.tensor([ms1, 0.0 if ms1 >= 0 else 1.0], device=dev, dtype=torch.float64)
        dist.all_reduce(t1, op=dist.ReduceOp.MAX)
        if float(t1[1]) > 0:
            wire = {'error': err1 or 'the one-rank cycle failed on another rank'}
        else:
            wire = {'dp_form_world1_ms_per_cycle_max_over_ranks': round(float(t1[0]), 3),
                    'us_per_average_wire_and_skew': round((dt / cycles * 1e3 - float(t1[0])) * 1e3 / 18, 2),
                    'design_budget': 'DESIGN section 6: 7.8x at w = 0, ~6.8x at w = 16 us, 6.0x at w = 30 us'}
        dist.barrier()
    # rollout alone (outside the timed region): policy forward + sampling + environment step, hipGraph-replayed collects
    tc = time.perf_counter()
    for _ in range(cycles if rollout_probe else 0):
        col.collect(steps_per_env * n_env, buf)
    torch.cuda.synchronize()
    t_collect = max(time.perf_counter() - tc, 1e-9)
    # ---- roofline leg (after the timed region): HIP event pairs on the directly launched GEMMs of the update half of a cycle (the
    # rollout half replays a hipGraph: its 2 products per vector step are not seen by the recorder and are counted analytically)
    roof = None
    nprof = 2
    from mansy_immersivevideostreaming_amd._lib import lib
    graph_mode = pol.graph_update
    pol.graph_update = False              # (the event pairs ride on direct launches: the profiled cycles run the update half un-captured)
    if rank == 0:
        ms_g, n_g, fl_g = _gemm_prof(lib(), cycle, nprof)
    else:
        for _ in range(nprof):
            cycle()
        torch.cuda.synchronize()
    pol.graph_update = graph_mode
    steps = world * n_env * steps_per_env * cycles
    if rank == 0:
        eps = steps / dt / world                                      # env-steps/s of this GPU
        # live: this library's launches per cycle = what the timed cycles enqueued directly + the captured rollout graph's kernels (one
        # replay per cycle); torch's own few kernels per cycle (uniforms, slab copies) from the committed profile, flagged if stale
        n_lib = (launches1 - launches0) / float(cycles) + (col.graph_launches if col.use_graph and col._graph is not None else 0)
        n_torch, launch_src, launch_stale = _ppo_torch_kernels_per_cycle()
        n_launch = round(n_lib + (n_torch or 0), 1)
        ppo_traffic, ppo_traffic_src, ppo_traffic_stale = _ppo_pmc_traffic()
        gemm_tf = fl_g / (ms_g * 1e-3) / 1e12 if ms_g > 0 else 0.0
        roof = {'bound': 'mfma', 'kernel': 'gemm_f32_dma_kernel (FeatureNet block-diagonal product, heads, dF / dW products of the update)',
                'achieved': round(eps * PPO_FLOP_PER_ENV_STEP / 1e12, 3), 'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                'frac': round(eps * PPO_FLOP_PER_ENV_STEP / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
                'traffic': ppo_traffic, 'traffic_unit': 'HBM-side bytes per env-step (whole cycle, PMC)', 'traffic_source': ppo_traffic_src,
                'traffic_stale': ppo_traffic_stale,
                'note': 'whole cycle on the as-written 11.5 MFLOP per env-step; the cycle is launch- and dependency-bound, not MFMA-bound',
                'hbm': {'achieved': round(eps * PPO_BYTES_PER_ENV_STEP / 1e9, 2), 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
                        'frac': round(eps * PPO_BYTES_PER_ENV_STEP / 1e9 / PEAK_HBM_GBS, 5)},
                'dominant_kernel_live': {'gemm_launches_per_cycle_update_half': n_g // nprof, 'avg_launch_us': round(ms_g * 1e3 / max(n_g, 1), 2),
                                         'gemm_ms_per_cycle': round(ms_g / nprof, 3), 'gemm_tflops': round(gemm_tf, 2),
                                         'gemm_frac_of_peak': round(gemm_tf / PEAK_F32_MFMA_TFLOPS, 4),
                                         'rollout_gemm_launches_per_cycle_in_graph': 2 * steps_per_env},
                'launch_floor': {'us_per_dependent_launch': LAUNCH_FLOOR_US, 'launches_per_cycle': n_launch,
                                 'library_launches_per_cycle_live': round(n_lib, 1), 'rollout_graph_kernels': col.graph_launches,
                                 'torch_kernels_per_cycle': n_torch, 'torch_kernels_source': launch_src, 'torch_kernels_stale': launch_stale,
                                 'floor_ms_per_cycle': round((n_launch or 0) * LAUNCH_FLOOR_US * 1e-3, 3),
                                 'measured_ms_per_cycle': round(dt / cycles * 1e3, 3)}}
    if peer:
        for ps in pol._peer.values():
            ps.check()                       # a timed-out wait would have poisoned the gradients with NaN: fail loudly
    spread = max(replica_spread(pol.engine.ac.flat_p, world), replica_spread(pol.engine.idn.flat_p, world))
    return {'metric': 'PPO env-steps/sec', 'value': round(steps / dt, 1), 'unit': 'env-steps/s', 'n_gpus': world, 'cycles': cycles,
            'timing': 'median of %d timed blocks of %d cycles each (a block = barrier + sync, cycles, sync + barrier; max over ranks per block)' % (PPO_BLOCKS, cycles),
            'replica_param_spread': spread, 'env_shards': shards, 'envs_per_gpu': n_env,
            'grad_sync': sync_desc, 'grad_sync_report': rep, 'per_rank': per_rank, 'wire_term': wire,
            'update_half': ('hipGraph replay' if pol.graph_replays else 'direct launches') + f' ({pol.graph_replays} replays so far)',
            'ms_per_cycle': round(dt / cycles * 1e3, 3), 'host_enqueue_ms_per_cycle': round(t_host / cycles * 1e3, 3),
            'update_graph_replays': pol.graph_replays, 'rollout_only_env_steps_per_s': round(n_env * steps_per_env * cycles / t_collect, 1),
            'rollout_step_latency_us': round(t_collect / (cycles * steps_per_env) * 1e6, 1), 'final_loss': float(np.mean(res['loss'])),
            'config': {'workload': f'{n_env} device-resident envs/GPU x {steps_per_env} steps per collect (4096 transitions/GPU), '
                                   'identifier train (2 full-batch rounds) + relabel + PPO update (minibatch 512, repeat 2), ' + (
                                       'REAL Jin2022 x 4G train tables of the reference (72-episode catalogue: 18 videos x 45 users x 24 traces x 4 preferences; '
                                       'tests/golden/env_tables_jin2022_4g.npz)' if tables_kind == 'real' else 'synthetic Jin2022/4G-shaped tables') + ', fp32',
                       'parallelism': f'dp{world}', 'tables': tables_kind},
            'data': 'real' if tables_kind == 'real' else 'synthetic',
            'model_flops_per_env_step': PPO_FLOP_PER_ENV_STEP, 'algorithmic_bytes_per_env_step': PPO_BYTES_PER_ENV_STEP, 'dtype': 'f32',
            'roofline': roof}


def bench_vp_inference(model, h, c, f, reps=5):
    """V10 + V11: `sample()` (KV-cached autoregressive decode, 3-head ensemble mean, wrap) and the tile-map / IoU metric of
    its output against the ground truth, as predict.py / the validation loop run them."""
    import torch
    from mansy_immersivevideostreaming_amd import kernels
    was_training = model.training
    model.eval()
    with torch.no_grad():
        pred = model.sample(h, c)
        torch.cuda.synchronize()
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        for _ in range(reps):
            pred = model.sample(h, c)
        e1.record()
        for _ in range(reps):
            g = kernels.tilemap(f.contiguous(), 2560, 1440, 8, 8)
            p = kernels.tilemap(pred.contiguous(), 2560, 1440, 8, 8)
            iou = kernels.tilemap_iou(g.reshape(-1), p.reshape(-1))
        e2.record()
        torch.cuda.synchronize()
    model.train(was_training)
    B, T = pred.shape[0], pred.shape[1]
    ms_s, ms_t = e0.elapsed_time(e1) / reps, e1.elapsed_time(e2) / reps
    return {'metric': 'viewport-trajectories/sec (VP sample)', 'value': round(B / ms_s * 1e3, 1), 'unit': 'trajectories/s', 'ms_per_call': round(ms_s, 3),
            'model_flops_per_trajectory': 0.174e9, 'tilemap_points_per_s': round(2 * B * T / ms_t * 1e3, 0), 'tilemap_ms': round(ms_t, 4),
            'mean_iou': float(iou.mean().item()),
            'config': {'workload': f'sample() B={B}, S=10, T={T}, d=512, eval mode + 2x{B * T} tile maps (8x8 tiles, 2560x1440) and IoU'}}


def bench_expert(dev, horizon=4, n_env=256, reps=20, cpu=True):
    """MPC expert: look-ahead decisions/s of the exhaustive 15^h plan search, all environments per launch,
    next to the sequential C oracle's literal scan on one host core."""
    import torch
    from mansy_immersivevideostreaming_amd.bitrate_selection.envs import expert_env as X
    T = X.EnvTables.synthetic(dev, seed=5, train_identifier_reward=False)
    venv = X.ExpertVecEnv(T, n_env, horizon, seed=0)
    venv.reset()
    for _ in range(3):
        venv.step(venv.choose_action())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        venv.choose_action()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    out = {'metric': 'MPC expert decisions/sec', 'value': round(n_env / ms * 1e3, 1), 'unit': 'decisions/s', 'ms_per_call': round(ms, 4),
           'plans_per_s': round(n_env * 15 ** horizon / ms * 1e3, 0),
           'config': {'workload': f'{n_env} environments x 15^{horizon} plans per decision (run_expert.py --horizon {horizon}), '
                                  'synthetic Jin2022/4G-shaped tables, f64 download integration + f32 QoE'}}
    if cpu:
        from oracle import env as oenv
        OT = oenv.EnvTables({k: T.host[k] for k in T.FIELDS}, T.host['qoe_w'], train_identifier_reward=False)
        ex = oenv.Expert(OT, venv.cache.vp_video.cpu().numpy(), horizon)
        oe = oenv.Env(OT, seed=0, worker_num=n_env)
        oe.reset()
        k, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < 1.0:
            ex.choose_action(oe)
            k += 1
        dt = time.perf_counter() - t0
        out['cpu_baseline'] = {'value': round(k / dt, 1), 'unit': 'decisions/s', 'cores': 1, 'kind': 'port',
                               'sample': f'{k} decisions of the C oracle (literal 15^{horizon} scan) in {dt:.1f}s'}
    return out


def cpu_baseline_ppo(seconds=6.0):
    """Reference-style rollout on one host core: sequential C-oracle env + B=1 oracle actor forward + sampling."""
    import numpy as np
    import torch
    from oracle import env as oenv
    from oracle import ppo_oracle as po
    nthreads = torch.get_num_threads()
    torch.set_num_threads(1)
    rs = np.random.RandomState(0)
    if os.path.exists(REAL_TABLES):       # the same tables as the GPU leg, host copy: the reference's Jin2022 x 4G train split
        z = np.load(REAL_TABLES)
        arrays = {k: z['train/' + k] for k in ('size', 'quality', 'video_len', 'vp_gt', 'vp_pred', 'vp_acc', 'vp_start', 'vp_end', 'trace_bw', 'trace_len', 'samples')}
        tables_desc = 'real Jin2022 x 4G train tables'
    else:
        arrays, tables_desc = _synthetic_arrays(), 'synthetic tables'
    OT = oenv.EnvTables(arrays, np.array([[7, 1, 1], [1, 7, 1], [1, 1, 7], [3, 3, 3]], np.float32), train_identifier_reward=True)
    env = oenv.Env(OT, seed=5, worker_num=1)
    sd = po.make_policy_state_dict(5)
    obs = env.reset()
    n, t0 = 0, time.time()
    row = np.zeros((1, 780), np.float32)
    with torch.no_grad():
        while time.time() - t0 < seconds:
            row[0, :779] = obs
            logits = po.actor_logits(sd, torch.from_numpy(row))
            a = int(po.categorical_sample(logits, torch.from_numpy(rs.rand(1).astype(np.float32)))[0])
            obs, r, done, _ = env.step(a)
            if done:
                obs = env.reset()
            n += 1
    dt = time.time() - t0
    torch.set_num_threads(nthreads)
    return {'value': round(n / dt, 1), 'unit': 'env-steps/s', 'cores': 1, 'kind': 'port',
            'sample': f'{n} sequential env steps (C oracle env + B=1 oracle actor forward + sampling) in {dt:.1f}s on the {tables_desc}'}


def _synthetic_arrays():
    """Host copy of EnvTables.synthetic's arrays (numpy only; no device)."""
    import numpy as np
    from mansy_immersivevideostreaming_amd.bitrate_selection.envs import mansy_env as me
    captured = {}

    class Cap(me.EnvTables):
        def __init__(self, arrays, qoe_weights, device, **kw):
            captured.update(arrays)
    Cap.synthetic('cpu', seed=5, n_sample=240)
    return captured


def spawn_ranks(n, argv, script=None):
    """`python bench.py --gpus N` without a launcher around it: this process (which has made no GPU call -- it has only
    compiled the library if it was missing) starts N fresh children of this very script, one per GPU, with the
    torch.distributed environment of a one-node job (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*), relays rank 0's JSON line and
    exits non-zero if any rank fails.  It never replaces itself (no exec): the children are ordinary subprocesses."""
    import socket
    import subprocess
    with socket.socket() as s:                    # a free rendezvous port on this node
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    base = dict(os.environ, WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                MANSY_BENCH_LAUNCHER='self-spawn')
    base.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    procs = []
    for r in range(n):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        # rank 0 inherits stdout (its JSON line IS this command's output); the other ranks print nothing there by contract, and
        # whatever they do print goes to stderr so it can never be mistaken for the result line
        procs.append(subprocess.Popen([sys.executable, script or os.path.abspath(__file__)] + argv, env=env, cwd=ROOT,
                                      stdout=None if r == 0 else sys.stderr))
    failed = None
    alive = set(range(n))
    grace = float(os.environ.get('MANSY_BENCH_EXIT_GRACE_S', '120'))
    first_exit = None
    while alive and failed is None:
        for r in sorted(alive):
            rc = procs[r].poll()
            if rc is None:
                continue
            alive.discard(r)
            if first_exit is None:
                first_exit = time.time()
            if rc != 0:
                failed = (r, rc)
                break
        # every rank ends with the same barrier + destroy_process_group: once one has exited 0 the others are seconds behind.  A rank
        # still running long after that is stuck (a hung destroy, a peer kernel that never met its peers): do not poll for ever.
        if failed is None and alive and first_exit is not None and time.time() - first_exit > grace:
            failed = (sorted(alive)[0], 124)
        time.sleep(0.05)
    if failed is not None:                        # one rank died: the others would wait in a collective for ever
        for r in alive:
            procs[r].terminate()
        deadline = time.time() + 10
        for r in alive:
            try:
                procs[r].wait(timeout=max(0.1, deadline - time.time()))
            except subprocess.TimeoutExpired:
                procs[r].kill()
        print(f'bench.py: rank {failed[0]} of {n} exited with code {failed[1]}', file=sys.stderr, flush=True)
        return failed[1] if failed[1] > 0 else 1
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=4096, help='trajectories per GPU')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    args = ap.parse_args()
    if args.gpus < 1:
        ap.error('--gpus must be >= 1')

    # a fresh clone has no in-tree library (*.so is git-ignored): local rank 0 compiles it, the others wait for the file
    from mansy_immersivevideostreaming_amd import build_ext
    if int(os.environ.get('LOCAL_RANK', '0')) == 0:
        build_ext.ensure_built()
    else:
        for _ in range(3600):
            if os.path.exists(build_ext.LIB):
                break
            time.sleep(0.5)

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))
    env_world_size = int(os.environ.get('WORLD_SIZE', '1'))
    if env_world_size != args.gpus:
        # a line whose n_gpus is not what was asked for would be read as a measurement of the wrong job
        sys.exit(f'bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={env_world_size} ranks; refusing to run')

    # before ANY torch.cuda call (device_count() can initialise HSA): the host driver only supports dmabuf IPC, and a torchrun launch
    # has no self-spawn parent that exported the variable
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    import numpy as np
    import torch
    import torch.distributed as dist
    from mansy_immersivevideostreaming_amd import dist as mdist
    # each rank's threads on a core set of their own (LOCAL_RANK-th slice of the allowed cores), before anything touches the GPU: at 8 ranks every
    # gradient average waits for the slowest rank's enqueue thread, and eight of them on one socket otherwise share whatever the scheduler gives
    host_cores = mdist.pin_host_cores() if env_world_size > 1 else None
    if os.environ.get('MANSY_XG_TEST_FAIL_RANK'):       # tests/test_gpu_dist.py: one rank's peer set-up fails (the package reads no environment: the harness injects)
        mdist.PeerGradSync._fail_setup_on_rank = int(os.environ['MANSY_XG_TEST_FAIL_RANK'])
    if env_world_size > 1 and os.environ.get('MANSY_SHARE_GPU') != '1' and torch.cuda.device_count() < env_world_size:
        sys.exit(f'bench.py: {env_world_size} ranks need {env_world_size} GPUs, this node shows {torch.cuda.device_count()} '
                 '(MANSY_SHARE_GPU=1 MANSY_DIST_BACKEND=gloo runs the functional test of the multi-rank path on fewer)')
    rank, world, local = mdist.init_process_group()
    dev = torch.device('cuda', local)
    torch.cuda.set_device(dev)
    # proof that the collective library saw every rank: its own world size and an all-reduce of one 1 per rank
    dist_info = {'backend': None, 'world_size': 1, 'ranks_reporting': 1,
                 'launcher': os.environ.get('MANSY_BENCH_LAUNCHER', 'torchrun' if 'TORCHELASTIC_RUN_ID' in os.environ else 'env')}
    if world > 1:
        ones = torch.ones(1, device=dev, dtype=torch.int32)
        dist.all_reduce(ones)
        dist_info.update(backend=dist.get_backend(), world_size=dist.get_world_size(), ranks_reporting=int(ones.item()))
        if dist_info['world_size'] != args.gpus or dist_info['ranks_reporting'] != args.gpus:
            sys.exit(f'bench.py: asked for {args.gpus} ranks, the process group reports {dist_info}')
        # self-diagnosis before any timed leg: devices, peer-access matrix, hipIpc open of a fine-grained allocation from every
        # peer + one peer-memory average in both forms, the library all-reduce, each rank's core set -- gathered on every rank, printed by rank 0
        dist_info['preflight'] = mdist.preflight(world, rank, dev, sizes=(1024, 435200))
        dist_info['preflight']['pinned_cores_this_rank'] = host_cores

    from mansy_immersivevideostreaming_amd.viewport_prediction.models import ViewportTransformerMTIO, FusedAdamW
    from mansy_immersivevideostreaming_amd._lib import lib, check

    B, S, T, d = args.batch, 10, 10, 512
    torch.manual_seed(5)
    random.seed(5)
    np.random.seed(5)
    model = ViewportTransformerMTIO(in_channel=2, fut_window=T, d_model=d, dim_feedforward=d, device=dev).to(dev)
    model.train()
    if world > 1:
        model.set_data_parallel(world)      # SyncBN for the DistillLayer: batch statistics over the GLOBAL mini-batch
    opt = FusedAdamW(model, lr=1e-4)
    h, c, f = (t.to(dev) for t in synthetic_trajectories(B, S, T, seed=5 + rank))

    # N > 1: flat-gradient averaging with the decoder-side two thirds of the all-reduce hidden under the encoder backward
    grad_sync = mdist.OverlappedGradSync(world, dev) if world > 1 else None

    def step():
        return model.train_step(h, c, f, opt, grad_sync=grad_sync)

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    loss_val = float(loss.item())
    value = world * B * args.steps / dt

    # ---- roofline leg (rank 0): a HIP event pair on every GEMM dispatch of the same workload
    roof = None
    # every rank runs these steps (the data-parallel step holds collectives: SyncBN statistics, gradient all-reduce -- a
    # rank-0-only step would leave the other ranks in a different collective); only rank 0 records GEMM launch times
    nprof = max(1, min(args.steps, 3))
    L = lib()
    # (the decoder recurrence runs as two half-batches on two streams in the timed legs; the per-kernel timing legs run it on one
    # stream, because concurrent kernels stretch each other's durations: a kernel's duration is then a property of the kernel)
    model.two_stream = False
    if rank == 0:
        ms, n, fl = _gemm_prof(L, step, nprof)
        achieved = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        traffic, traffic_src, traffic_stale = _pmc_traffic('r*_pmc_gemm.json')
        roof = {'bound': 'mfma', 'kernel': 'gemm_f32_dma_kernel (v_mfma_f32_32x32x2_f32, LDS-DMA staged)', 'achieved': round(achieved, 2),
                'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s', 'frac': round(achieved / PEAK_F32_MFMA_TFLOPS, 4),
                'traffic': traffic, 'traffic_source': traffic_src, 'traffic_stale': traffic_stale,
                'launches_per_step': n // nprof, 'avg_launch_us': round(ms * 1e3 / max(n, 1), 2),
                'gemm_flops_per_step': fl / nprof, 'gemm_ms_per_step': round(ms / nprof, 3),
                'algorithmic_flops_per_step': FLOP_PER_TRAJ * B,
                'model_frac': round(value / world * FLOP_PER_TRAJ / 1e12 / PEAK_F32_MFMA_TFLOPS, 4)}
    else:
        for _ in range(nprof):
            step()
        torch.cuda.synchronize()

    # ---- the same step in the split-bf16 precision modes (secondary lines; every rank runs them: collectives inside)
    modes = []
    for mode, nprod in (('bf16x3', 3), ('bf16x6', 6), ('bf16', 1)):      # bf16: ONE product (a perf mode: errors ~1e-3, the class of the reference's TF32 setting)
        model.precision = mode
        model.two_stream = None
        for _ in range(2):
            step()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        tm = time.perf_counter()
        for _ in range(args.steps):
            mloss = step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dtm = time.perf_counter() - tm
        if world > 1:
            t = torch.tensor([dtm], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dtm = t.item()
        model.two_stream = False
        if rank == 0:
            ms, n, fl = _gemm_prof(L, step, nprof)
            alg_tf = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
            traffic, traffic_src, traffic_stale = _pmc_traffic(f'r*_pmc_gemm_{mode}.json')
            vm = world * B * args.steps / dtm
            modes.append({'dtype': mode, 'metric': 'viewport-trajectories/sec (VP train)', 'value': round(vm, 1), 'unit': 'trajectories/s',
                          'ms_per_step': round(dtm / args.steps * 1e3, 3), 'final_loss': float(mloss.item()), 'speedup_vs_f32': round(vm / value, 3),
                          'roofline': {'bound': 'mfma', 'kernel': f'gemm_bf16{{f,g,h,p,s}}_kernel (v_mfma_f32_32x32x16_bf16, {nprod} bf16 products per fp32 product; weights pre-split, '
                                                                  'activations split at fragment read / in the staging pass)',
                                       # executed bf16 MFMA FLOPs = nprod x the algorithmic (fp32-product) FLOPs
                                       'achieved': round(alg_tf * nprod, 2), 'peak': PEAK_BF16_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                                       'frac': round(alg_tf * nprod / PEAK_BF16_MFMA_TFLOPS, 4), 'traffic': traffic, 'traffic_source': traffic_src,
                                       'traffic_stale': traffic_stale,
                                       'algorithmic_tflops': round(alg_tf, 2), 'launches_per_step': n // nprof,
                                       'avg_launch_us': round(ms * 1e3 / max(n, 1), 2), 'gemm_ms_per_step': round(ms / nprof, 3),
                                       'model_frac_of_f32_peak': round(vm / world * FLOP_PER_TRAJ / 1e12 / PEAK_F32_MFMA_TFLOPS, 4)}})
        else:
            for _ in range(nprof):
                step()
            torch.cuda.synchronize()
    model.precision = None
    model.two_stream = None

    vp_spread = replica_spread(model._flat_p, world)
    ppo = bench_ppo(rank, world, dev, mdist, cycles=max(2, min(args.steps, 6)), warmup=3)      # warm-up: direct, capture + replay, replay

    if rank == 0:
        out = {
            'metric': 'viewport-trajectories/sec (VP train)', 'value': round(value, 1), 'unit': 'trajectories/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': f'VP Transformer train step (fwd+loss+bwd+AdamW), B={B}/GPU synthetic torus-walk '
                                   f'trajectories len 21 (hist 10 + cur 1 + pred 10), d=512, 8 heads, 2+2 layers, dropout on, '
                                   f'fp32 MFMA, decoder recurrence as two half-batches on two streams', 'two_stream': True, 'global_batch': B * world, 'parallelism': f'dp{world}'},
            'final_loss': loss_val, 'replica_param_spread': vp_spread, 'dist': dist_info,
            'roofline': roof,
            'precision_modes': modes,
            'secondary': ppo,
        }
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline()
            out['secondary']['cpu_baseline'] = cpu_baseline_ppo()
        if world == 1:      # single-GPU extras (no collectives inside, but keep every rank's tail identical at N > 1)
            out['tertiary'] = bench_expert(dev, cpu=not args.no_cpu_baseline)
            out['inference'] = bench_vp_inference(model, h, c, f)
            out['small_batch'] = bench_vp_small(dev)
            out['configs4'] = bench_ppo_c5(dev)
            # the same cycle on the synthetic bench-shaped tables every earlier round's line was measured on (the real-table line above should sit within a few % of it)
            ms_syn, nl_syn, _ = _ppo_cycle_time(_ppo_policy(dev), dev, max(2, min(args.steps, 6)), 2)
            out['secondary']['synthetic_tables'] = {'value': round(256 * 16 / ms_syn * 1e3, 1), 'unit': 'env-steps/s', 'ms_per_cycle': round(ms_syn, 3),
                                                    'library_launches_per_cycle': round(nl_syn, 1), 'data': 'synthetic',
                                                    'real_over_synthetic': round(out['secondary']['value'] / (256 * 16 / ms_syn * 1e3), 4)}
            out['secondary']['dp_form'] = bench_ppo_dp_form(dev, mdist, vp_leg=lambda: vp_dp_form_leg(model, opt, h, c, f, dev, mdist))      # (last: it brings up a one-rank RCCL group)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
