"""Step-by-step probe of PPOPolicy.update against oracle.ppo_oracle.update on the GPU box (what located the value-clip gradient
flip documented in tests/test_gpu_ppo.py::test_whole_update_vs_oracle_update): per minibatch step the loss / value-loss difference, the worst
weight difference, the engine's gradient norm, and around the diverging steps the per-tensor gradient norms of both sides.
    python tools/ppo_update_probe.py"""
import sys, os, numpy as np, torch, ctypes
R=os.environ.get('GRAFT_REPO_ROOT','/root/repo'); sys.path.insert(0, R); sys.path.insert(0, R+'/tests')
import test_gpu_ppo as t
from oracle import ppo_oracle as po
from mansy_immersivevideostreaming_amd._lib import check, lib, ptr, stream_ptr
class NS: pass
from mansy_immersivevideostreaming_amd.bitrate_selection.models import mansy, mansy_ppo
from mansy_immersivevideostreaming_amd.bitrate_selection.envs import mansy_env
M = NS(); M.mansy, M.ppo, M.env = mansy, mansy_ppo, mansy_env
Z = t.Z
sd = po.make_policy_state_dict(int(Z['wseed']))
T, N, bs = 16, 256, 512
pol = t.build_policy(M, sd)
rs = np.random.RandomState(3); n = T * N; src = Z['obs']
obs = src[rs.randint(0, len(src), size=n)].reshape(T, N, 780).copy()
obs_next = src[rs.randint(0, len(src), size=n)].reshape(T, N, 780).copy()
act = rs.randint(0, 15, size=(T, N)).astype(np.int32); rew = rs.randn(T, N).astype(np.float32); done = rs.rand(T, N) < 0.05
buf = M.ppo.RolloutBuffer(T, N, 'cuda'); rms_o, ost = po.RunningMeanStd(), {}
gsnaps = []
_orig_clip = torch.nn.utils.clip_grad_norm_
def _rec(params, max_norm):
    gsnaps.append([p.grad.detach().clone() for p in params]); return _orig_clip(params, max_norm)
torch.nn.utils.clip_grad_norm_ = _rec
for it in range(2):
    gsnaps.clear()
    r_it = rew + 0.1 * it
    buf.obs.copy_(torch.from_numpy(obs)); buf.obs_next.copy_(torch.from_numpy(obs_next)); buf.act.copy_(torch.from_numpy(act))
    buf.rew.copy_(torch.from_numpy(r_it)); buf.done.copy_(torch.from_numpy(done.astype(np.uint8))); buf.filled = T
    snaps = []
    np.random.seed(100 + it)
    want, inter = po.update({k: v.clone() for k, v in sd.items()}, obs, obs_next, act, r_it, done, rms_o, ost, lamb=0.5, batch_size=bs, repeat=2,
                            on_step=lambda k, u: snaps.append({n_: v.detach().clone() for n_, v in u.items()}))
    np.random.seed(100 + it)
    pol.relabel(buf, 0.5)
    data = pol.process_fn(buf)
    for key in ('returns', 'adv', 'v_s', 'logp_old'):
        print(it, key, 'max diff', float(np.abs(data[key].cpu().numpy() - inter[key]).max()))
    eng, f = pol.engine, pol.engine.ac
    k = 0
    for rep in range(2):
        chunks = list(mansy_ppo.split_indices(n, bs))
        for chunk in chunks:
            idx = torch.from_numpy(chunk.astype(np.int32)).cuda()
            f.step += 1
            stats = torch.zeros(4, device='cuda')
            arr, garr = f.pointers(grads=True)
            check(lib().mansy_ppo_minibatch_step(arr, garr, ptr(f.flat_p), ptr(f.flat_g), ptr(f.m), ptr(f.v), f.flat_p.numel(), ptr(data['obs']),
                                                 ptr(idx), ptr(data['act']), ptr(data['adv']), ptr(data['logp_old']), ptr(data['v_s']),
                                                 ptr(data['returns']), idx.numel(), 0.2, 0.5, 0.02, 1, 1, 0.0, 1.0, 5e-4, 1e-2, f.step, -1, 0, ptr(stats),
                                                 ptr(eng.workspace()), eng.max_batch, 0, None, 0, None, None, eng.prec, stream_ptr()), 'mb')
            torch.cuda.synchronize()
            worst = ('', 0.0)
            for name, o, p in zip([x[0] for x in f.table], f.offsets, f.params):
                e = float(np.abs(f.flat_p[o:o + p.numel()].view(p.shape).cpu().numpy() - snaps[k][name].numpy()).max())
                if e > worst[1]: worst = (name, e)
            g = f.flat_g.double().norm().item()
            if it == 1 and k in (9, 10, 11):
                names = list(ost['uniq'].keys())
                for name, o, p in zip([x[0] for x in f.table], f.offsets, f.params):
                    ge = f.flat_g[o:o + p.numel()].view(p.shape).cpu().numpy(); go = gsnaps[k][names.index(name)].numpy()
                    print('      grad', name, 'engine norm %.5f oracle norm %.5f maxdiff %.2e' % (np.linalg.norm(ge), np.linalg.norm(go), np.abs(ge - go).max()))
            print(it, k, 'loss diff %.2e vf diff %.2e' % (abs(stats[0].item() - want[k][0]), abs(stats[2].item() - want[k][2])), 'worst weight', worst, 'gnorm(engine) %.4f' % g)
            k += 1
