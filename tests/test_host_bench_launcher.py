"""bench.py's own launcher (`python bench.py --gpus N` with no torchrun around it): N fresh rank processes with the
torch.distributed environment, rank 0's stdout relayed, non-zero exit if any rank fails, refusal to run when the launcher's
world size is not --gpus.  CPU-only: the children here are small stand-in scripts, not the benchmark."""
import json
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _script(tmp_path, body):
    p = tmp_path / 'child.py'
    p.write_text(textwrap.dedent(body))
    return str(p)


def test_spawn_ranks_environment_and_gloo_rendezvous(tmp_path, capfd):
    import bench
    child = _script(tmp_path, '''
        import json, os, sys
        import torch, torch.distributed as dist
        dist.init_process_group('gloo')
        t = torch.ones(1)
        dist.all_reduce(t)
        rec = dict(rank=dist.get_rank(), world=dist.get_world_size(), heads=int(t.item()), local=os.environ['LOCAL_RANK'],
                   addr=os.environ['MASTER_ADDR'], launcher=os.environ['MANSY_BENCH_LAUNCHER'], argv=sys.argv[1:])
        open(sys.argv[1] + '.%d' % dist.get_rank(), 'w').write(json.dumps(rec))
        print('{"metric": "from rank %d"}' % dist.get_rank(), flush=True)
        dist.destroy_process_group()
    ''')
    rc = bench.spawn_ranks(2, [str(tmp_path / 'out'), '--gpus', '2'], script=child)
    assert rc == 0
    recs = [json.load(open(str(tmp_path / 'out') + '.%d' % r)) for r in range(2)]
    for r, rec in enumerate(recs):
        assert rec['rank'] == r and rec['world'] == 2 and rec['heads'] == 2 and rec['local'] == str(r)
        assert rec['addr'] == '127.0.0.1' and rec['launcher'] == 'self-spawn' and rec['argv'][1:] == ['--gpus', '2']
    out, err = capfd.readouterr()
    # rank 0's line is the command's stdout; the other rank's stdout was diverted to stderr
    assert out.count('{"metric"') == 1 and 'from rank 0' in out and 'from rank 1' in err


def test_spawn_ranks_fails_when_a_rank_fails(tmp_path):
    import bench
    child = _script(tmp_path, '''
        import os, sys, time
        if os.environ['RANK'] == '1':
            sys.exit(7)
        time.sleep(600)        # a rank waiting in a collective for the dead one
    ''')
    import time
    t0 = time.time()
    rc = bench.spawn_ranks(3, [], script=child)
    assert rc == 7 and time.time() - t0 < 60


def test_bench_refuses_a_world_size_other_than_gpus():
    env = dict(os.environ, WORLD_SIZE='3', RANK='0', LOCAL_RANK='1')      # LOCAL_RANK 1: do not compile, just wait for the file
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1'], env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and 'refusing to run' in r.stderr and '{"metric"' not in r.stdout
