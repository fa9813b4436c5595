"""Build libtante_hip.so (gfx950) in-tree with hipcc.  No torch dependency: the library is a plain
C-ABI shared object (include/tante_hip.h)."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libtante_hip.so")
SOURCES = ["gemm.hip", "attention.hip", "pointwise.hip", "block_fused.hip", "block_sliced.hip", "block_bwd.hip", "block_bwd_fs.hip", "train.hip", "backward.hip", "wgrad.hip", "head_fused.hip", "head_enc.hip", "operators.hip", "enc_fused.hip", "axis_bwd.hip", "spectral_dft.hip", "cvit_fused.hip", "axis_mfma.hip", "tail_chain.hip"]
HEADERS = [os.path.join(CSRC, "common.hip.h"), os.path.join(CSRC, "fused_common.hip.h"), os.path.join(CSRC, "fs_common.hip.h"), os.path.join(CSRC, "block_sliced.h"), os.path.join(CSRC, "spectral_dft.h"), os.path.join(os.path.dirname(HERE), "include", "tante_hip.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=fast"]
# the fused bf16 kernels are bound by VALU issue: without NaN-honouring every fmaxf / clamp loses its v_max canonicalisation
EXTRA_FLAGS = {"block_fused.hip": ["-fno-honor-nans", "-DTANTE_MFMA_SETPRIO", "-mllvm", "-amdgpu-sched-strategy=max-ilp"], "block_sliced.hip": ["-fno-honor-nans", "-DTANTE_MFMA_SETPRIO", "-DFS_PRIO=1"], "block_bwd.hip": ["-fno-honor-nans", "-DTANTE_MFMA_SETPRIO", "-DBT_PRIO"], "block_bwd_fs.hip": ["-fno-honor-nans", "-DTANTE_MFMA_SETPRIO", "-DBT_PRIO"], "head_fused.hip": ["-fno-honor-nans"], "head_enc.hip": ["-fno-honor-nans"], "enc_fused.hip": ["-fno-honor-nans"], "operators.hip": ["-fno-honor-nans"], "spectral_dft.hip": ["-fno-honor-nans"], "cvit_fused.hip": ["-fno-honor-nans", "-DTANTE_MFMA_SETPRIO"], "axis_mfma.hip": ["-fno-honor-nans"], "tail_chain.hip": ["-fno-honor-nans"]}


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False, ablate: bool = False) -> str:
    """ablate=True builds tools/_ab/libtante_ablate.so with -DTANTE_ABLATE: the only build in which the TANTE_*_DEBUG
    timing-ablation switches (which produce wrong results on purpose) read the environment.  The product library never does."""
    libdir = os.path.join(os.path.dirname(HERE), "tools", "_ab", "ablate_obj") if ablate else LIBDIR
    lib = os.path.join(os.path.dirname(HERE), "tools", "_ab", "libtante_ablate.so") if ablate else LIB
    os.makedirs(libdir, exist_ok=True)
    objs, jobs = [], []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        op = os.path.join(libdir, src.replace(".hip", ".o"))
        objs.append(op)
        if force or _stale(op, [sp] + HEADERS):
            jobs.append([HIPCC, *FLAGS, *(["-DTANTE_ABLATE"] if ablate else []), *EXTRA_FLAGS.get(src, []), "-c", sp, "-o", op])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed: {' '.join(cmd)}\n{r.stdout}\n{r.stderr}")

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(run, jobs))
    if jobs or force or _stale(lib, objs):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs, "-L/opt/rocm/lib", "-lhipfft", "-Wl,-rpath,/opt/rocm/lib"])
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, ablate="--ablate" in sys.argv))
