"""Builds libmansy_hip.so (gfx950) in-tree with hipcc.  No torch dependency, no JIT cache:
the .so lives next to the sources so it travels with the repo snapshot to the GPU box."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OBJ = os.path.join(CSRC, '_obj')
LIB = os.path.join(HERE, 'libmansy_hip.so')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-fno-gpu-rdc', '-Wno-unused-result']


def sources(lab=False):
    """The library's translation units; lab=True adds csrc/lab/*.hip (the -DMANSY_LAB twin's settable default variant: never in the release build)."""
    out = sorted(f for f in os.listdir(CSRC) if f.endswith('.hip'))
    if lab:
        out += sorted(os.path.join('lab', f) for f in os.listdir(os.path.join(CSRC, 'lab')) if f.endswith('.hip'))
    return out


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


STAMP = LIB + '.stamp'


def source_digest():
    """sha256 over every source / header the library is built from (+ the flags): what ensure_built() compares, because file
    times do not survive the copy to the GPU box."""
    import hashlib
    h = hashlib.sha256(' '.join(FLAGS).encode())
    files = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(('.hip', '.h'))]
    files.append(os.path.join(os.path.dirname(HERE), 'include', 'mansy_hip.h'))
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, 'rb') as fh:
            h.update(fh.read())
    return h.hexdigest()


GEMM_SOURCES = ('gemm_f32.hip', 'gemm_bf16s.hip', 'gemm_bf16a.hip', 'gemm_tile.h', 'gemm_wsk.h', 'mansy_kernels.h', 'mansy_common.h')


def gemm_source_digest():
    """sha256 over the sources of the GEMM kernels only (+ the flags): stamped into profiles/*_pmc_gemm*.json when the PMC passes
    are aggregated (tools/pmc_aggregate.py) and compared by bench.py, so that a `roofline.traffic` taken from counters of an
    older kernel is flagged (`traffic_stale`)."""
    import hashlib
    h = hashlib.sha256(' '.join(FLAGS).encode())
    for f in GEMM_SOURCES:
        h.update(f.encode())
        with open(os.path.join(CSRC, f), 'rb') as fh:
            h.update(fh.read())
    return h.hexdigest()


LAB_LIB = os.path.join(HERE, 'libmansy_hip_lab.so')


def build(force=False, verbose=False, lab=False):
    """lab=True: the same sources with -DMANSY_LAB -> libmansy_hip_lab.so, the only build that has a settable default kernel-selection
    variant (mansy_lab_set_variant) for whole-engine A/B timings (tools/) and the paired-launch equivalence test.  The release library
    (lab=False) holds no process-wide mutable state."""
    if lab:
        return _build(os.path.join(CSRC, '_obj_lab'), LAB_LIB, FLAGS + ['-DMANSY_LAB'], force, verbose, stamp=None, lab=True)
    return _build(OBJ, LIB, FLAGS, force, verbose, stamp=STAMP)


def _build(OBJ, LIB, FLAGS, force, verbose, stamp, lab=False):
    os.makedirs(OBJ, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]
    headers.append(os.path.join(os.path.dirname(HERE), 'include', 'mansy_hip.h'))
    jobs = []
    objs = []
    for s in sources(lab):
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ, os.path.basename(s)[:-4] + '.o')
        objs.append(obj)
        if force or _stale(obj, [src] + headers):
            jobs.append([HIPCC] + FLAGS + ['-c', src, '-o', obj])

    def run(cmd):
        if verbose:
            print(' '.join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('hipcc failed: %s\n%s\n%s' % (' '.join(cmd), r.stdout, r.stderr))
        return r
    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _stale(LIB, objs):
        run([HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs)
    if stamp:
        with open(stamp, 'w') as fh:
            fh.write(source_digest())
    return LIB


def ensure_built():
    """The in-tree library, (re)compiled if this checkout has none or if it was built from other sources than the ones in the
    tree now (a fresh clone: *.so is git-ignored; an edit of csrc/ or include/mansy_hip.h: the digest in the stamp file no
    longer matches).  Not a fallback -- it builds the HIP path; without hipcc it raises.  _lib.lib() additionally checks
    mansy_abi_version() against its prototypes."""
    def fresh():
        if os.path.exists(LIB) and os.path.exists(STAMP):
            with open(STAMP) as fh:
                return fh.read().strip() == source_digest()
        return False
    if fresh():
        return LIB
    # several ranks of one node may get here at once (torchrun on a fresh clone): one builds, the others wait for it
    import fcntl
    with open(LIB + '.lock', 'w') as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            return LIB if fresh() else build()
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True, lab='--lab' in sys.argv))
