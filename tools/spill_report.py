"""Register / scratch report of every gfx950 kernel in spartan2_amd/lib/*.o: the .hip_fatbin section of each object is copied out (llvm-objcopy), the
gfx950 code object unbundled (clang-offload-bundler) and its AMDGPU metadata notes read (llvm-readelf --notes): .vgpr_count, .agpr_count,
.vgpr_spill_count, .sgpr_spill_count, .private_segment_fixed_size (scratch bytes per lane), .wavefront... Runs on a CPU-only box.
usage: python tools/spill_report.py [--md out.md] [--spills-only]
tests/test_spills_cpu.py imports kernels() and fails when a kernel on its allow-list of hot names spills."""
import argparse
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = os.environ.get("SPARTAN_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"
FIELDS = (".vgpr_count", ".agpr_count", ".sgpr_count", ".vgpr_spill_count", ".sgpr_spill_count", ".private_segment_fixed_size", ".group_segment_fixed_size", ".max_flat_workgroup_size")


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout.splitlines()
        return out if len(out) == len(names) else names
    except Exception:
        return names


def kernels(lib_dir=None):
    """[{object, name (demangled), vgpr_count, ..., vgpr_spill_count, private_segment_fixed_size}] for every kernel of every object"""
    lib_dir = lib_dir or os.path.join(ROOT, "spartan2_amd", "lib")
    rows = []
    with tempfile.TemporaryDirectory() as tmp:
        for obj in sorted(glob.glob(os.path.join(lib_dir, "*.o"))):
            fat, co = os.path.join(tmp, "f.fatbin"), os.path.join(tmp, "f.co")
            subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", obj, fat], check=True)
            if not os.path.exists(fat) or os.path.getsize(fat) == 0:
                continue
            subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={fat}", f"--targets={TARGET}", f"--output={co}"], check=True)
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True, check=True).stdout
            cur = None
            for line in notes.splitlines():
                m = re.match(r"\s*(-\s+)?(\.[a-z_]+):\s*(.*)$", line)
                if not m:
                    continue
                dash, key, val = m.group(1), m.group(2), m.group(3).strip()
                if dash and key in (".agpr_count", ".args"):  # a kernel's map starts with its alphabetically first key
                    cur = {"object": os.path.basename(obj)}
                    rows.append(cur)
                if cur is None:
                    continue
                if key == ".name":
                    cur["mangled"] = val.strip("'\"")
                elif key in FIELDS:
                    try:
                        cur[key[1:]] = int(val)
                    except ValueError:
                        pass
            os.remove(fat)
            os.remove(co)
    rows = [r for r in rows if "mangled" in r and "vgpr_count" in r]
    for r, n in zip(rows, demangle([r["mangled"] for r in rows])):
        r["name"] = n
    return rows


def short(name, width=110):
    name = re.sub(r"\(.*$", "", name)  # drop the parameter list
    return name if len(name) <= width else name[: width - 3] + "..."


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--md")
    ap.add_argument("--spills-only", action="store_true")
    a = ap.parse_args()
    rows = kernels()
    rows.sort(key=lambda r: (-r.get("vgpr_spill_count", 0), -r.get("private_segment_fixed_size", 0), r["object"], r["name"]))
    lines = ["| object | kernel | VGPR | AGPR | VGPR spills | SGPR spills | scratch B/lane | LDS B |", "|---|---|---|---|---|---|---|---|"]
    for r in rows:
        if a.spills_only and not (r.get("vgpr_spill_count", 0) or r.get("private_segment_fixed_size", 0)):
            continue
        lines.append(f"| {r['object']} | `{short(r['name'])}` | {r.get('vgpr_count', '')} | {r.get('agpr_count', '')} | {r.get('vgpr_spill_count', 0)} | {r.get('sgpr_spill_count', 0)} | "
                     f"{r.get('private_segment_fixed_size', 0)} | {r.get('group_segment_fixed_size', 0)} |")
    text = "\n".join(lines)
    nsp = sum(1 for r in rows if r.get("vgpr_spill_count", 0))
    text += f"\n\n{len(rows)} kernels, {nsp} with VGPR spills, {sum(1 for r in rows if r.get('private_segment_fixed_size', 0))} with a private segment (scratch)\n"
    if a.md:
        with open(a.md, "w") as f:
            f.write(text)
    print(text)
    return 0


if __name__ == "__main__":
    sys.exit(main())
