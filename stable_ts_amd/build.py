"""Build libswx.so (hipcc, gfx950) in-tree.  `python -m stable_ts_amd.build`"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libswx.so")
SOURCES = ["swx_runtime.hip", "swx_gemm.hip", "swx_norm.hip", "swx_attn.hip", "swx_decode.hip", "swx_align.hip",
           "swx_mel.hip", "swx_dtw.hip", "swx_decstep.hip", "swx_loudness.hip", "swx_headsel.hip", "swx_flac.hip"]


def _newest(paths):
    return max(os.path.getmtime(p) for p in paths)


def build_variant(name: str, defines, verbose: bool = False) -> str:
    """An experiment build of the whole library with extra -D switches, written to scripts/exp/libswx_<name>.so (git-ignored,
    shipped to the GPU box); the A/B scripts copy it over stable_ts_amd/libswx.so on the box.  Never loaded by the product."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    outdir = os.path.join(os.path.dirname(HERE), "scripts", "exp")
    objdir = os.path.join(outdir, "_build_" + name)
    os.makedirs(objdir, exist_ok=True)
    procs, objs = [], []
    for s in SOURCES:
        o = os.path.join(objdir, s + ".o")
        objs.append(o)
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", *[f"-D{d}" for d in defines],
               "-c", os.path.join(CSRC, s), "-o", o]
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {s}:\n{out.decode()}")
    out = os.path.join(outdir, f"libswx_{name}.so")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
    return out


def build(force: bool = False, verbose: bool = True) -> str:
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    deps = srcs + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + \
        [os.path.join(os.path.dirname(HERE), "include", "swx.h")]
    if not force and os.path.exists(OUT) and os.path.getmtime(OUT) >= _newest(deps):
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(HERE, "_build")
    os.makedirs(objdir, exist_ok=True)
    objs = []
    procs = []
    for s in srcs:
        o = os.path.join(objdir, os.path.basename(s) + ".o")
        objs.append(o)
        if not force and os.path.exists(o) and os.path.getmtime(o) >= _newest([s] + deps[len(srcs):]):
            continue
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-c", s, "-o", o]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    failed = False
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed = True
            sys.stderr.write(f"--- {s}\n{out.decode()}\n")
        elif verbose and out.strip():
            print(out.decode())
    if failed:
        raise RuntimeError("hipcc failed")
    tmp = OUT + f".{os.getpid()}.tmp"          # link beside the target, then rename: a process that is loading the
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + objs      # library never sees a half-written file
    if verbose:
        print(" ".join(cmd), flush=True)
    try:
        subprocess.check_call(cmd)
        # a shared library links with unresolved symbols without complaint: load it once before it replaces the good one
        probe = subprocess.run([sys.executable, "-c", f"import ctypes; ctypes.CDLL({tmp!r})"], capture_output=True, text=True)
        if probe.returncode != 0:
            raise RuntimeError("libswx.so does not load: " + probe.stderr.strip().splitlines()[-1])
        os.replace(tmp, OUT)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
