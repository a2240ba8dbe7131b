"""Builds libmuse_hip.so (the C-ABI shared library, include/muse_hip.h) for gfx950 with hipcc, in-tree.

hipcc cross-compiles without a GPU; the .so is git-ignored but travels to the GPU box with the tree."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libmuse_hip.so')
SOURCES = ['gemm.hip', 'gemm_big.hip', 'gemm_pers.hip', 'gemm_cfg.hip', 'gemm_wide.hip', 'gemm_wide_conv.hip', 'gemm_terms.hip', 'gemm_fp8.hip', 'gemm_tn.hip', 'fp8_act.hip', 'attention.hip', 'cross_fold.hip', 'cross_vw_x2.hip', 'attention_bwd.hip', 'train.hip', 'train_step.hip', 'train_prep.hip', 'norm_act.hip', 'sampling.hip', 'sampling_fused.hip', 'vae.hip', 'vq.hip', 'parity.hip', 'split.hip', 'attention_f32.hip', 'attention_x2.hip', 'vae_model.hip', 'model.hip', 'comm.hip', 'api.hip']
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function',
         '-fno-fast-math', '-ffp-contract=off']


def _headers():
    return [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')] + [os.path.join(HERE, '..', 'include', 'muse_hip.h'), __file__]


def _flags():
    flags = list(FLAGS)
    if os.environ.get('MM_GEMM_ABLATE'):      # tools/gemm_bench.py ablations only: run-time skip-stores / -DMA / -MFMA switches in the k-loops
        flags.append('-DMM_GEMM_ABLATE')
    if os.environ.get('MM_GEMM_TIMING'):      # tools/cfg2_timing.py only: cycle stamps inside gemm_cfg2_kernel
        flags.append('-DMM_GEMM_TIMING')
    return flags


def _obj(src):
    return os.path.join(HERE, 'build', src.replace('.hip', '.o'))


def _stale_sources(flags):
    """the translation units whose object is older than the source, any header, this file, or was built with other flags"""
    stamp = os.path.join(HERE, 'build', 'flags.txt')
    same_flags = os.path.exists(stamp) and open(stamp).read() == ' '.join(flags)
    hdr = max(os.path.getmtime(h) for h in _headers())
    out = []
    for src in SOURCES:
        o = _obj(src)
        if not same_flags or not os.path.exists(o) or os.path.getmtime(o) < max(hdr, os.path.getmtime(os.path.join(CSRC, src))):
            out.append(src)
    return out


def build(force=False, verbose=False, jobs=None):
    flags = _flags()
    os.makedirs(os.path.join(HERE, 'build'), exist_ok=True)
    todo = list(SOURCES) if force else _stale_sources(flags)
    objs = [_obj(s) for s in SOURCES]
    if not todo and os.path.exists(LIB) and os.path.getmtime(LIB) >= max(os.path.getmtime(o) for o in objs):
        return LIB
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    jobs = jobs or max(2, min(16, (os.cpu_count() or 8)))
    failed = False
    pending = list(todo)
    running = []
    while pending or running:
        while pending and len(running) < jobs:
            src = pending.pop(0)
            cmd = [hipcc, *flags, '-c', os.path.join(CSRC, src), '-o', _obj(src)]
            if verbose:
                print(' '.join(cmd))
            running.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        src, p = running.pop(0)
        out, _ = p.communicate()
        if p.returncode != 0:
            failed = True
            sys.stderr.write(f'--- {src} failed ---\n{out}\n')
        elif verbose and out.strip():
            print(out)
    if failed:
        raise RuntimeError('hipcc failed')
    with open(os.path.join(HERE, 'build', 'flags.txt'), 'w') as f:
        f.write(' '.join(flags))
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB, *objs, '-ldl']
    subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
