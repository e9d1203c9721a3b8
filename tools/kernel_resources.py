#!/usr/bin/env python3
"""Per-kernel resource usage of the built library: VGPRs, AGPRs, SGPRs, spills, scratch, LDS, code size -- read from the
gfx950 code object's metadata notes (no GPU needed).

    python tools/kernel_resources.py [path/to/libafg_hip.so] [--match substring] [--json out.json]
"""
import argparse
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def code_objects(lib, tmp):
    """Unbundle every gfx950 code object of `lib` (one per translation unit) into tmp; return their paths."""
    out = subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--list", "--type=o", f"--input={lib}"], capture_output=True, text=True)
    paths = []
    # a shared library holds several bundles back to back (one __CLANG_OFFLOAD_BUNDLE__ per object): split by magic
    data = open(lib, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    starts = [m.start() for m in re.finditer(re.escape(magic), data)]
    for n, s in enumerate(starts):
        e = starts[n + 1] if n + 1 < len(starts) else len(data)
        blob = os.path.join(tmp, f"bundle{n}.bin")
        with open(blob, "wb") as fh:
            fh.write(data[s:e])
        co = os.path.join(tmp, f"co{n}.elf")
        r = subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={blob}",
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], capture_output=True, text=True)
        if r.returncode == 0 and os.path.exists(co) and os.path.getsize(co) > 0:
            paths.append(co)
    return paths


def kernels_of(co):
    txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True).stdout
    rows, cur = [], None
    for line in txt.splitlines():
        m = re.match(r"\s*-? ?\.(\w+):\s*(.*)$", line.strip())
        if not m:
            continue
        k, v = m.group(1), m.group(2).strip().strip("'")
        if k == "agpr_count" or (k == "args" and cur is None):
            pass
        if line.strip().startswith("- .agpr_count") or (line.strip().startswith("- .") and k in ("agpr_count", "args")):
            cur = {}
            rows.append(cur)
        if cur is not None and k in ("name", "vgpr_count", "agpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count",
                                     "private_segment_fixed_size", "group_segment_fixed_size", "max_flat_workgroup_size"):
            cur[k] = v if k == "name" else int(v)
    rows = [r for r in rows if "name" in r and "vgpr_count" in r]
    return rows


def demangle(names):
    r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return r.stdout.splitlines()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("lib", nargs="?", default=os.path.join(ROOT, "audio-formats_amd", "lib", "libafg_hip.so"))
    ap.add_argument("--match", default="")
    ap.add_argument("--json", default="")
    args = ap.parse_args()
    with tempfile.TemporaryDirectory() as tmp:
        rows = []
        for co in code_objects(args.lib, tmp):
            rows += kernels_of(co)
    names = demangle([r["name"] for r in rows])
    for r, n in zip(rows, names):
        n = re.sub(r"^void ", "", n).replace("(anonymous namespace)::", "")
        depth, cut = 0, len(n)
        for i, ch in enumerate(n):                     # the argument list opens at the first '(' outside template brackets
            if ch == "<":
                depth += 1
            elif ch == ">":
                depth -= 1
            elif ch == "(" and depth == 0:
                cut = i
                break
        r["kernel"] = n[:cut]
    rows = [r for r in rows if args.match in r["kernel"]]
    rows.sort(key=lambda r: r["kernel"])
    print(f"{'kernel':78s} {'vgpr':>4s} {'agpr':>4s} {'sgpr':>4s} {'vspill':>6s} {'scratch':>7s} {'lds':>6s}")
    for r in rows:
        print(f"{r['kernel'][:78]:78s} {r.get('vgpr_count', 0):4d} {r.get('agpr_count', 0):4d} {r.get('sgpr_count', 0):4d} "
              f"{r.get('vgpr_spill_count', 0):6d} {r.get('private_segment_fixed_size', 0):7d} {r.get('group_segment_fixed_size', 0):6d}")
    if args.json:
        with open(args.json, "w") as fh:
            json.dump([{k: v for k, v in r.items() if k != "name"} for r in rows], fh, indent=1)


if __name__ == "__main__":
    sys.exit(main())
