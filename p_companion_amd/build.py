"""Builds libpcompanion_hip.so (gfx950) in-tree with hipcc.  No torch involved: the library
is a plain C-ABI shared object (include/pcompanion_hip.h)."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libpcompanion_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function"]
FLAGS += os.environ.get("PC_EXTRA_HIPCC_FLAGS", "").split()      # developer builds (e.g. -DPC_NT_TIMING, scripts/nt_phase_times.py)


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp")))


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(HERE, "..", "include", "pcompanion_hip.h"))
    jobs = []
    objs = []
    for s in sources():
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ, os.path.splitext(s)[0] + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + headers):
            cmd = [HIPCC] + FLAGS + (["-x", "hip"] if s.endswith(".hip") else []) + ["-c", src, "-o", obj]
            jobs.append(cmd)

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        if verbose and r.stderr.strip():
            print(r.stderr)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    if jobs or force or _stale(LIB, objs):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


LLVM_BIN = os.environ.get("PC_LLVM_BIN", "/opt/rocm/lib/llvm/bin")


def kernel_resources():
    """Per kernel of every compiled translation unit: the code object's own metadata (llvm-readelf --notes on the gfx950 image
    unbundled from the object's .hip_fatbin): {name: {vgpr_count, agpr_count, sgpr_count, vgpr_spill_count, sgpr_spill_count,
    private_segment_fixed_size, lds (group_segment_fixed_size), unit}}.  Used by __graft_entry__.build() to refuse a build in
    which a product kernel spills vector registers to scratch."""
    import tempfile
    try:
        import yaml
    except ImportError as e:
        raise RuntimeError("spill gate unavailable: PyYAML missing (the code objects' metadata note is YAML)") from e
    for tool in ("llvm-objcopy", "clang-offload-bundler", "llvm-readelf"):
        if not os.path.exists(os.path.join(LLVM_BIN, tool)):
            raise RuntimeError(f"spill gate unavailable: {tool} missing under {LLVM_BIN}")
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        for s in sources():
            if not s.endswith(".hip"):
                continue
            unit = os.path.splitext(s)[0]
            obj = os.path.join(OBJ, unit + ".o")
            fat, co = os.path.join(tmp, unit + ".fat"), os.path.join(tmp, unit + ".co")
            r = subprocess.run([os.path.join(LLVM_BIN, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, obj],
                               capture_output=True, text=True)
            if r.returncode != 0 or not os.path.exists(fat):
                if "hip_fatbin" in (r.stderr or "") or not os.path.exists(obj):
                    continue                                      # (a unit without device code / not built: no section to dump)
                raise RuntimeError(f"spill gate: llvm-objcopy failed on {obj}: {(r.stderr or '').strip()[:300]}")
            subprocess.run([os.path.join(LLVM_BIN, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fat,
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], check=True, capture_output=True)
            notes = subprocess.run([os.path.join(LLVM_BIN, "llvm-readelf"), "--notes", co], check=True, capture_output=True,
                                   text=True).stdout
            # the note is YAML between '---' and '...'
            lines = notes.splitlines()
            lo = next(i for i, ln in enumerate(lines) if ln.strip() == "---")
            hi = next((i for i in range(lo + 1, len(lines)) if lines[i].strip() == "..."), len(lines))
            meta = yaml.safe_load("\n".join(lines[lo + 1:hi]))
            for k in meta.get("amdhsa.kernels", []):
                out[k[".name"]] = {"unit": unit, "lds": int(k.get(".group_segment_fixed_size", 0)),
                                   **{f: int(k.get("." + f, 0)) for f in ("vgpr_count", "agpr_count", "sgpr_count", "vgpr_spill_count",
                                                                          "sgpr_spill_count", "private_segment_fixed_size")}}
    return out


def spilling_kernels():
    """[(name, vgpr_spill_count, private_segment_fixed_size)] of kernels that keep vector registers in scratch."""
    return sorted((n, r.get("vgpr_spill_count", 0), r.get("private_segment_fixed_size", 0)) for n, r in kernel_resources().items()
                  if r.get("vgpr_spill_count", 0) > 0 or r.get("private_segment_fixed_size", 0) > 0)


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    if "--resources" in sys.argv:
        for n, r in sorted(kernel_resources().items()):
            print(r["unit"], n, {k: v for k, v in r.items() if k != "unit"})
    bad = spilling_kernels()
    print("kernels with scratch:", bad if bad else "none")
