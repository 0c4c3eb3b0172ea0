#!/usr/bin/env python3
"""Static audit of gfx950 ISA (`hipcc -S`) for MFMA-result hazards that the compiler left uncovered along ANY control-flow path.

Why (r04): the r03 streaming-convolution bug ("wrong, run-to-run different 16-bit inference") was NOT a vmcnt problem.  In the
kernel with MUBUF stores hipcc (ROCm 7.2, clang 22) scheduled the last `v_mfma_f32_16x16x32_bf16` of an output row directly in
front of a wave-uniform branch (`if (want_sums)`), and the block at the branch target begins with `v_accvgpr_read_b32` of that
MFMA's result: on the TAKEN edge (no statistics pointer) only 2 wait states separate the two, the hazard recogniser's `s_nop`
sits behind that first read (it covers the following three reads, which have exactly the 8 states it gives every other row), and
the read returns the accumulator's previous contents — element 3 of every lane's 4-channel group, in every output row of that loop
phase.  tests/diag/stream_race_diag.py shows exactly that footprint (rows y = 3 mod 4, channels 19, 23, 27, 31), and
tests/diag/vmcnt_order_probe.hip shows that vmcnt retires in issue order (stores never overtake older loads).

What is checked: for every MFMA, along every path of the function's control-flow graph (conditional branches: both edges), the
number of wait states (instructions issued; `s_nop N` = N + 1) up to the first NON-MFMA instruction that names a register of
the MFMA's destination (read or overwrite).  Required states: the NEED table below, calibrated on the toolchain's own padding of
straight-line code.  MFMA consumers taking the whole destination as SrcC are the accumulate chain (no wait); MFMAs reading it as
SrcA/SrcB are reported separately.

    python tools/mfma_hazard_audit.py file.s|lib.so [...]        exit code 1 if any violation is found
(a .so is taken apart with llvm-objcopy / llvm-objdump: the audit then sees the code that ships, not a recompilation)
"""
import re
import sys

# mnemonic prefix -> required wait states between the MFMA and a non-MFMA access of its destination.  Calibrated on this toolchain's
# own padding of straight-line code (the minimum over ~200 sites per shape in conv_halo.hip): 16-bit XDL shapes passes + 4
# (16x16x32: 4 passes -> 8), f32-input shapes passes + 2 (16x16x4_f32: 8 passes -> 10).
NEED = {
    "v_mfma_f32_16x16x32": 8, "v_mfma_f32_16x16x16": 8, "v_mfma_f32_32x32x16": 12, "v_mfma_f32_32x32x8": 12,
    "v_mfma_f32_16x16x4_f32": 10, "v_mfma_f32_32x32x2_f32": 18, "v_mfma_f32_4x4x4": 6,
}
REG = re.compile(r"\b([av])(?:\[(\d+):(\d+)\]|(\d+)\b)")
LABEL = re.compile(r"^([.A-Za-z_][\w.$]*):")


def regs_of(tok):
    out = set()
    for m in REG.finditer(tok):
        kind = m.group(1)
        if m.group(2) is not None:
            lo, hi = int(m.group(2)), int(m.group(3))
        else:
            lo = hi = int(m.group(4))
        for r in range(lo, hi + 1):
            out.add((kind, r))
    return out


OBJ_FUNC = re.compile(r"^[0-9a-f]+ <([^>]+)>:")


def parse_functions(path):
    """Both forms: the compiler's assembly (`hipcc -S`: `name:` / `.LBBn_m:` labels) and `llvm-objdump -d --symbolize-operands`
    of a code object (`addr <name>:` / `addr <Ln>:` labels, `// addr: encoding` comments)."""
    funcs = {}
    cur = None
    for raw in open(path, errors="replace"):
        mo = OBJ_FUNC.match(raw)
        if mo:
            name = mo.group(1)
            if re.fullmatch(r"L\d+", name):
                if cur is not None:
                    funcs[cur]["labels"][name] = len(funcs[cur]["ins"])
            else:
                cur = name
                funcs[cur] = {"ins": [], "labels": {}}
            continue
        line = raw.split("//")[0].split(";")[0].rstrip()
        m = LABEL.match(line)
        if m:
            name = m.group(1)
            if not name.startswith(".L"):
                cur = name
                funcs[cur] = {"ins": [], "labels": {}}
            elif cur is not None:
                funcs[cur]["labels"][name] = len(funcs[cur]["ins"])
            continue
        s = line.strip()
        if not s or s.startswith(".") or cur is None or not raw.startswith(("\t", " ")):
            continue
        funcs[cur]["ins"].append(s)
    return {k: v for k, v in funcs.items() if any(i.startswith("v_mfma") for i in v["ins"])}


def disassemble_library(so_path, out_dir):
    """The gfx950 code objects inside a HIP shared library (one clang offload bundle per translation unit in `.hip_fatbin`),
    disassembled with llvm-objdump; returns the list of disassembly files."""
    import os
    import struct
    import subprocess
    llvm = os.environ.get("VK_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
    fat = os.path.join(out_dir, "fat.bin")
    subprocess.run([f"{llvm}/llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", so_path, os.path.join(out_dir, "stripped.tmp")], check=True)
    b = open(fat, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    outs = []
    pos = b.find(magic)
    while pos >= 0:
        (ne,) = struct.unpack_from("<Q", b, pos + 24)
        q = pos + 32
        for _ in range(ne):
            off, size, ts = struct.unpack_from("<QQQ", b, q)
            q += 24
            triple = b[q:q + ts].decode()
            q += ts
            if "gfx950" in triple and size:
                co = os.path.join(out_dir, f"co_{len(outs)}.co")
                open(co, "wb").write(b[pos + off:pos + off + size])
                dis = co[:-3] + ".dis"
                with open(dis, "w") as fh:
                    subprocess.run([f"{llvm}/llvm-objdump", "-d", "--symbolize-operands", co], stdout=fh, check=True)
                outs.append(dis)
        pos = b.find(magic, pos + 1)
    return outs


def states(ins):
    m = re.match(r"s_nop\s+(\d+)", ins)
    return int(m.group(1)) + 1 if m else 1


def audit(path):
    bad = []
    info = []
    for fname, f in parse_functions(path).items():
        ins, labels = f["ins"], f["labels"]
        n = len(ins)
        for i, s in enumerate(ins):
            if not s.startswith("v_mfma"):
                continue
            mn = s.split()[0]
            need = next((p for k, p in NEED.items() if mn.startswith(k)), None)
            if need is None:
                info.append(f"{path}:{fname}: unknown MFMA shape {mn}")
                continue
            ops = s[len(mn):].split(",")
            dest = regs_of(ops[0])
            # DFS over paths: (index, states so far)
            stack = [(i + 1, 0, ())]
            seen = {}
            while stack:
                j, d, trail = stack.pop()
                if j >= n or d >= need or len(trail) > 64:
                    continue
                if seen.get(j, 1 << 30) <= d:
                    continue
                seen[j] = d
                t = ins[j]
                tm = t.split()[0]
                if tm.startswith("v_mfma"):
                    tops = t[len(tm):].split(",")
                    tdest, ta, tb, tc = regs_of(tops[0]), regs_of(tops[1]), regs_of(tops[2]), regs_of(tops[3]) if len(tops) > 3 else set()
                    if (ta | tb) & dest:
                        info.append(f"{path}:{fname}: MFMA result used as SrcA/B of `{t}` after {d} states (line {j})")
                        continue
                    if tc & dest or tdest & dest:
                        continue                      # accumulate chain / overwritten by the next MFMA: the matrix pipe orders these
                    stack.append((j + 1, d + 1, trail))
                    continue
                if tm in ("s_endpgm",):
                    continue
                if tm == "s_branch":
                    tgt = t.split()[1]
                    if tgt in labels:
                        stack.append((labels[tgt], d + 1, trail + (j,)))
                    continue
                if tm.startswith("s_cbranch"):
                    tgt = t.split()[1]
                    if tgt in labels:
                        stack.append((labels[tgt], d + 1, trail + (j,)))
                    stack.append((j + 1, d + 1, trail))
                    continue
                if regs_of(t[len(tm):]) & dest:
                    bad.append((path, fname, i, s, j, t, d, need, [ins[k] for k in trail]))
                    continue
                stack.append((j + 1, d + states(t), trail))
    return bad, info


def scratch_report(co_files):
    """[(kernel, scratch bytes, vgprs, spilled vgprs, spilled sgprs)] for every kernel of the code objects that uses scratch memory or spills
    (llvm-readelf --notes: the AMDGPU metadata).  r04: an edit that pushed a 250-register tile kernel over its cap spilled 11
    registers into the epilogue without any diagnostic — the CPU suite now looks."""
    import os
    import subprocess
    llvm = os.environ.get("VK_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
    out = []
    pat = re.compile(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)\n(?:.*\n)*?\s+\.sgpr_spill_count:\s+(\d+)\n(?:.*\n)*?"
                     r"\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)")
    for co in co_files:
        notes = subprocess.run([f"{llvm}/llvm-readelf", "--notes", co], capture_output=True, text=True, check=True).stdout
        for m in pat.finditer(notes):
            if int(m.group(2)) or int(m.group(5)):
                out.append((m.group(1), int(m.group(2)), int(m.group(4)), int(m.group(5)), int(m.group(3))))
    return out


def main():
    args = sys.argv[1:]
    if not args:
        print(__doc__)
        return 2
    files = []
    tmp = None
    for a in args:
        if a.endswith(".so"):
            import tempfile
            tmp = tmp or tempfile.mkdtemp(prefix="vk_audit_")
            d = tempfile.mkdtemp(dir=tmp)
            files += disassemble_library(a, d)
        else:
            files.append(a)
    total = 0
    nk = 0
    for p in files:
        bad, info = audit(p)
        nk += len(parse_functions(p))
        for (path, fname, i, s, j, t, d, need, trail) in bad:
            total += 1
            print(f"VIOLATION {path}\n  kernel {fname}\n  [{i}] {s}\n  [{j}] {t}\n  wait states on this path: {d}, required {need}"
                  + (f"\n  via taken branches: {trail}" if trail else ""))
        for x in sorted(set(info)):
            print("info:", x)
    for f in files:
        co = f[:-4] + ".co"
        import os
        if f.endswith(".dis") and os.path.exists(co):
            for name, scratch, vgprs, spills, sspills in scratch_report([co]):
                # scalar registers parked in VGPR lanes (v_writelane / v_readlane) leave their frame slots in the private-segment size
                # although no instruction touches scratch memory: reported apart from real scratch use
                if spills == 0 and sspills > 0 and scratch <= 4 * sspills:
                    print(f"sgpr-spill: {name} {sspills} SGPRs in VGPR lanes (private_segment {scratch} B unused), {vgprs} VGPRs")
                else:
                    print(f"scratch: {name} private_segment {scratch} B, {vgprs} VGPRs, {spills} spilled")
    print(f"{len(files)} file(s), {nk} kernels with MFMAs: {total} violation(s)")
    if tmp:
        import shutil
        shutil.rmtree(tmp, ignore_errors=True)
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main())
