"""Builds libmuse_hip.so (the C-ABI shared library, include/muse_hip.h) for gfx950 with hipcc, in-tree.

hipcc cross-compiles without a GPU; the .so is git-ignored but travels to the GPU box with the tree."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libmuse_hip.so')
SOURCES = ['gemm.hip', 'gemm_big.hip', 'gemm_pers.hip', 'gemm_cfg.hip', 'gemm_wide.hip', 'gemm_terms.hip', 'gemm_pp.hip', 'gemm_fp8.hip', 'fp8_act.hip', 'attention.hip', 'cross_fold.hip', 'attention_bwd.hip', 'train.hip', 'train_step.hip', 'norm_act.hip', 'sampling.hip', 'sampling_fused.hip', 'vae.hip', 'vq.hip', 'parity.hip', 'split.hip', 'attention_f32.hip', 'attention_x2.hip', 'vae_model.hip', 'model.hip', 'comm.hip', 'api.hip']
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function',
         '-fno-fast-math', '-ffp-contract=off']


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, '..', 'include', 'muse_hip.h'), __file__]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    flags = list(FLAGS)
    if os.environ.get('MM_GEMM_ABLATE'):      # tools/gemm_bench.py ablations only: run-time skip-stores / -DMA / -MFMA switches in the k-loops
        flags.append('-DMM_GEMM_ABLATE')
    if os.environ.get('MM_GEMM_TIMING'):      # tools/cfg2_timing.py only: cycle stamps inside gemm_cfg2_kernel
        flags.append('-DMM_GEMM_TIMING')
    objs = []
    procs = []
    os.makedirs(os.path.join(HERE, 'build'), exist_ok=True)
    for src in SOURCES:
        obj = os.path.join(HERE, 'build', src.replace('.hip', '.o'))
        objs.append(obj)
        cmd = [hipcc, *flags, '-c', os.path.join(CSRC, src), '-o', obj]
        if verbose:
            print(' '.join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    failed = False
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed = True
            sys.stderr.write(f'--- {src} failed ---\n{out}\n')
        elif verbose and out.strip():
            print(out)
    if failed:
        raise RuntimeError('hipcc failed')
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB, *objs, '-ldl']
    subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
