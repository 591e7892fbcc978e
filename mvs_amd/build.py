"""Build recipe for libmvs_hip.so (hipcc, gfx950 only, in-tree).

    python -m mvs_amd.build            # incremental
    python -m mvs_amd.build --force

The library is built next to the sources (mvs_amd/csrc/libmvs_hip.so) so it
travels with the repo snapshot to the GPU box; it is git-ignored.
"""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "obj")
LIB = os.path.join(CSRC, "libmvs_hip.so")
SOURCES = ("capi", "sweep", "sweep_persist", "regress", "conv3d_direct", "conv3d_mfma", "costreg", "conv_bf16x6", "conv_f16x3", "conv_f16x3_y8p", "conv_split", "conv_s2_march", "deconv_split", "tail_fused", "conv2d_pair", "conv2d_mfma", "feature_head", "fpn_tail", "conv3d_wgrad", "conv3d_wgrad_f16", "conv2d_wgrad", "bnorm", "cas_hypo", "geo_filter", "cvp_hypo", "cvp_glue", "imgprep", "fusibile", "camera")
# experiments kept with their tests, compiled into the tuning build only (dead weight in the release .so)
TUNING_SOURCES = ("conv_f16x3_pairs", "conv_f16x3_y8")
HEADERS = (os.path.join(CSRC, "mvs_common.h"), os.path.join(CSRC, "sweep_common.h"), os.path.join(CSRC, "conv_persistent.h"), os.path.join(CSRC, "conv_split_common.h"), os.path.join(CSRC, "conv_guard.h"), os.path.join(CSRC, "split2.h"),
           os.path.join(os.path.dirname(HERE), "include", "mvs_hip_tuning.h"), os.path.join(os.path.dirname(HERE), "include", "mvs_hip.h"))
# -ffp-contract=off: the plane-sweep coordinate arithmetic places its FMAs by
# hand to match the reference bit for bit; everything else uses fmaf/MFMA.
# -fno-slp-vectorize: v_pk_fma_f32 retires two results in 5-7 cycles (scripts/micro/pk_fma.hip), no
# faster than two v_fma_f32, and the vectoriser pays for its pairs with register moves (85 per tap
# iteration of the prob kernel): prob 0.36 -> 0.33 ms, conv9 0.145 -> 0.134 ms without it.
# -fvisibility=hidden: only what include/mvs_hip.h (and, tuning build, mvs_hip_tuning.h) declares leaves the library (the
# mvs:: launchers leaked as 64 extra text symbols)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-ffp-contract=off",
         "-fno-fast-math", "-fno-slp-vectorize", "-Wall", "-Wno-unused-function"]


# sources compiled WITH the SLP vectoriser (v_pk_fma_f32 / v_pk_mul_f32 pairs); MVS_BUILD_SLP="a,b" overrides for experiments
SLP_SOURCES = tuple(x for x in os.environ.get("MVS_BUILD_SLP", "").split(",") if x)


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libmvs_hip.so cannot be built on this machine")
    return exe


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, tuning=False):
    """tuning: libmvs_hip_tuning.so with -DMVS_TUNING -- the ablation / phase-stamp switches (MVS_CONV_SPLIT_ABL,
    MVS_CONV_SPLIT_LAPS, MVS_PROB_ABL, MVS_HEAD_ABL, MVS_SWEEP_ABLATE, the flag fields of MVS_SWEEP_PERSIST) exist only
    there: several of them give wrong results by design and write cycle counters through a caller's tensor, so the
    release library does not read them.  MVS_HIP_TUNING=1 makes mvs_amd._lib load that build."""
    obj_dir = OBJ + ("_tuning" if tuning else "")
    lib = LIB.replace(".so", "_tuning.so") if tuning else LIB
    os.makedirs(obj_dir, exist_ok=True)
    cc = hipcc()
    jobs = []
    sources = SOURCES + (TUNING_SOURCES if tuning else ())
    for name in sources:
        src = os.path.join(CSRC, name + ".hip")
        obj = os.path.join(obj_dir, name + ".o")
        if force or _stale(obj, (src,) + HEADERS):
            flags = [f for f in FLAGS if not (f == "-fno-slp-vectorize" and name in SLP_SOURCES)]
            jobs.append([cc] + flags + (["-DMVS_TUNING"] if tuning else []) + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        return r.stderr

    with ThreadPoolExecutor(max_workers=max(1, min(len(jobs), os.cpu_count() or 1))) as ex:
        for warn in ex.map(run, jobs):
            if verbose and warn.strip():
                print(warn)
    objs = [os.path.join(obj_dir, n + ".o") for n in sources]
    if force or jobs or _stale(lib, objs):
        run([cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
    return lib


ABI_TEST_SRC = os.path.join(os.path.dirname(HERE), "tests", "cpp", "abi_chain.cpp")
ABI_TEST_BIN = os.path.join(os.path.dirname(HERE), "tests", "cpp", "abi_chain")


def build_abi_test(force=False):
    """The standalone C++ caller of include/mvs_hip.h (tests/cpp/abi_chain.cpp), linked against
    the in-tree libmvs_hip.so; like the library it is built here and travels with the snapshot."""
    if not os.path.exists(ABI_TEST_SRC):
        return None
    if force or _stale(ABI_TEST_BIN, (ABI_TEST_SRC, LIB, HEADERS[-1])):
        cmd = [hipcc(), "--offload-arch=gfx950", "-O2", "-std=c++17", "-I" + os.path.join(os.path.dirname(HERE), "include"),
               ABI_TEST_SRC, "-L" + CSRC, "-lmvs_hip", "-Wl,-rpath,$ORIGIN/../../mvs_amd/csrc", "-o", ABI_TEST_BIN]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
    return ABI_TEST_BIN


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, tuning="--tuning" in sys.argv))
    print(build_abi_test(force="--force" in sys.argv))
