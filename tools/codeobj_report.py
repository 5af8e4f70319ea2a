#!/usr/bin/env python3
"""Resource usage of the built gfx950 code object (libvvcx.so): per kernel the metadata the loader sees (VGPRs, spills, LDS, scratch) and instruction counts from the
disassembly (scratch_ / buffer_ / global_ / flat_ / ds_ accesses, v_dot*, v_mfma*, s_barrier).  Run after `make -C <pkg>/csrc`:

    python tools/codeobj_report.py [--out profiles/r03_codeobj.json]

The CPU suite reads the same numbers (tests/test_host_cpu.py::test_resource_budget_of_the_compress_kernel): the stream kernel is sized for four workgroups per CU
(<= 128 VGPRs, <= 40 KB LDS); one byte of LDS too many silently costs a quarter of the resident streams, and the compiler then also drops the register target."""
import argparse, json, os, re, shutil, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "reduce-complexity-for-intra-coding-of-vvc_amd", "libvvcx.so")
LLVM = "/opt/rocm/lib/llvm/bin"


def extract(so, tmp):
    fat, co = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "k.co")
    subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, so, os.path.join(tmp, "copy.so")])
    subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + fat, "--output=" + co, "--unbundle"])
    return co


def kernel_metadata(co):
    """amdhsa.kernels of the code object's notes: one YAML list item per kernel (its first key behind "  - ", the others at the same depth, nested lists such as
    .args deeper).  The keys of an item are collected first and filed under the item's own .name (the keys are sorted, so half of them precede .name)."""
    notes = subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "--notes", co]).decode()
    keep = ("vgpr_count", "agpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "group_segment_fixed_size", "private_segment_fixed_size", "max_flat_workgroup_size")
    items, cur, inside = [], None, False
    for line in notes.split("\n"):
        if line.startswith("amdhsa.kernels:"):
            inside = True
            continue
        if inside and line and not line.startswith(" "):       # next top-level key of the metadata document
            inside = False
        if not inside:
            continue
        m = re.match(r"^  - \.(\w+):\s*(.*)$", line)
        if m:
            cur = {}
            items.append(cur)
        else:
            m = re.match(r"^    \.(\w+):\s*(.*)$", line)
        if m and cur is not None:
            cur[m.group(1)] = m.group(2).strip().strip("'")
    out = {}
    for it in items:
        name = it.get("name", "")
        if name.startswith("vvcx_") and name.endswith(("_u8", "_u16", "_kernel")):
            out[name] = {k: int(it[k]) for k in keep if k in it}
    return out


def instruction_mix(co):
    asm = subprocess.check_output([os.path.join(LLVM, "llvm-objdump"), "-d", "--mcpu=gfx950", co]).decode().split("\n")
    mix = {k: 0 for k in ("scratch_", "buffer_", "global_", "flat_", "ds_", "v_dot", "v_mfma", "s_barrier", "s_waitcnt", "total")}
    for l in asm:
        t = l.strip().split("//")[0].strip()
        if not t or t.endswith(":") or re.match(r"^[0-9a-f]+ <", t):
            continue
        mix["total"] += 1
        for k in mix:
            if k != "total" and t.startswith(k):
                mix[k] += 1
    return mix


def report(so=SO):
    tmp = tempfile.mkdtemp()
    try:
        co = extract(so, tmp)
        return {"library": os.path.relpath(so, ROOT), "kernels": kernel_metadata(co), "instructions": instruction_mix(co)}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--lib", default=SO)
    a = ap.parse_args()
    r = report(a.lib)
    txt = json.dumps(r, indent=1, sort_keys=True)
    if a.out:
        open(a.out, "w").write(txt + "\n")
    k = r["kernels"].get("vvcx_compress_kernel_u8", {})
    print("vvcx_compress_kernel_u8:", k)
    print("instructions:", r["instructions"])
