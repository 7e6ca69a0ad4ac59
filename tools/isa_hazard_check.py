#!/usr/bin/env python3
"""CPU-side check of the shipped gfx950 code objects: is any MFMA destination read (or overwritten) by a non-MFMA instruction closer behind the
matrix instruction than the hardware needs?

Why: round 4's first cut of k_mimi_rowlin returned an accumulator without its last product in one register of lanes 48..63, deterministically, where
hipcc had hoisted a v_accvgpr_read to right behind the end of an MFMA chain.  Round 5 measured what the chip needs (tools/probes/mfma_hazard, MI355X,
ROCm 7.2): between a v_mfma_f32_16x16x32_bf16 and the first VALU read of its destination 8 wait states (registers 2, 3 of the tile; 7 for 0, 1), 12 for
v_mfma_f32_32x32x16_bf16 (its last registers), the same for VGPR and AGPR destinations; ONE independent MFMA issued in between leaves 4, two leave 0.
Those are the numbers of LLVM's gfx950 hazard table (passes + 4), so a compiler-scheduled read is safe -- what the table cannot see is an instruction
inside an inline-asm statement (cdna_hip_programming.md 5.7 item 2), and five shipped kernels mix inline asm with MFMA chains.  This scan needs no GPU.

Model (conservative): a wait state per instruction issued, N + 1 for `s_nop N`; an intervening MFMA counts 4 (measured: >= 4 for either shape); the
scan follows straight-line code and stops at a branch, a barrier, s_waitcnt with a vmcnt/lgkmcnt (hundreds of cycles in practice) or after 16 states.
An MFMA that takes the destination as its C operand (the accumulation chain) is not a hazard (0 states, hardware-interlocked).

A second scan looks for the hazard the failing cut actually hit -- a packed-f32 op with a mixed operand selection right in front of an MFMA: scan_packed.

usage: isa_hazard_check.py libptts_hip.so                        (exit 1 on any violation)
       isa_hazard_check.py --asm file.s                          (hipcc -S --cuda-device-only output)
"""
import os
import re
import struct
import subprocess
import sys
import tempfile

LLVM_BIN = "/opt/rocm/lib/llvm/bin"
NEED = {"16x16x32": 8, "32x32x16": 12, "16x16x16": 8, "32x32x8": 12, "32x32x2": 12, "16x16x4": 8, "4x4x4": 6}   # wait states D -> non-MFMA read / write
MFMA_BETWEEN = 4
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(path):
    """Every gfx950 code object (ELF bytes) bundled in a host object / shared library's .hip_fatbin section."""
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, "fat.bin")
        subprocess.run([os.path.join(LLVM_BIN, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", path, fat], check=True)
        blob = open(fat, "rb").read()
    out = []
    pos = 0
    while True:
        pos = blob.find(MAGIC, pos)
        if pos < 0:
            break
        n = struct.unpack_from("<Q", blob, pos + 24)[0]
        p = pos + 32
        for _ in range(n):
            off, size, tsz = struct.unpack_from("<QQQ", blob, p)
            triple = blob[p + 24:p + 24 + tsz].decode()
            p += 24 + tsz
            if "gfx950" in triple and size:
                out.append(blob[pos + off:pos + off + size])
        pos += 24
    return out


def disassemble(elf_bytes):
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(elf_bytes)
        f.flush()
        r = subprocess.run([os.path.join(LLVM_BIN, "llvm-objdump"), "-d", "--mcpu=gfx950", f.name], capture_output=True, text=True, check=True)
    return r.stdout


REG = re.compile(r"\b([av])\[(\d+):(\d+)\]|\b([av])(\d+)\b")


def regs_of(operand):
    s = set()
    for m in REG.finditer(operand):
        if m.group(1):
            s.update((m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1))
        else:
            s.add((m.group(4), int(m.group(5))))
    return s


def parse(text):
    """-> {kernel: [(mnemonic, [operands])]} from llvm-objdump -d output or a .s file."""
    kernels, cur = {}, None
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]* ?<?([A-Za-z_][\w.$]*)>?:\s*(;.*)?$", line.strip()) if not line.startswith(("\t", " ")) else None
        if m and not line.strip().startswith((".", ";")):
            cur = kernels.setdefault(m.group(1), [])
            continue
        if cur is None:
            continue
        body = line.split("//")[0].split(";")[0].strip()
        if not body or body.startswith(".") or body.endswith(":"):
            continue
        parts = body.split(None, 1)
        mn = parts[0]
        ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
        cur.append((mn, ops))
    return kernels


def shape_of(mn):
    m = re.search(r"_(\d+x\d+x\d+)", mn)
    return m.group(1) if m else None


def scan(kernels, min_report=None):
    """-> (violations, closest) where closest[(kernel, shape)] = the smallest distance seen."""
    bad, closest = [], {}
    for k, ins in kernels.items():
        for i, (mn, ops) in enumerate(ins):
            if not mn.startswith(("v_mfma", "v_smfmac")) or not ops:
                continue
            shape = shape_of(mn)
            need = NEED.get(shape, 12)
            dst = regs_of(ops[0])
            states = 0
            for j in range(i + 1, min(i + 40, len(ins))):
                m2, o2 = ins[j]
                if m2.startswith(("s_cbranch", "s_branch", "s_barrier", "s_endpgm", "s_setpc", "s_swappc")):
                    break
                if m2 == "s_waitcnt" and any(("vmcnt" in o or "lgkmcnt" in o) for o in o2):
                    break
                if m2.startswith(("v_mfma", "v_smfmac")):
                    srcs = set().union(*[regs_of(o) for o in o2[1:3]]) if len(o2) >= 3 else set()
                    if srcs & dst:                      # D as an A / B operand of a later MFMA: the same requirement as a VALU read
                        hit = True
                    elif regs_of(o2[0]) & dst:          # accumulate chain (or a full overwrite by the next chain): interlocked
                        break
                    else:
                        states += MFMA_BETWEEN
                        if states >= 16:
                            break
                        continue
                else:
                    touched = set().union(*[regs_of(o) for o in o2]) if o2 else set()
                    hit = bool(touched & dst)
                if hit:
                    key = (k, shape)
                    if key not in closest or states < closest[key][0]:
                        closest[key] = (states, i, j, m2)
                    if states < need:
                        bad.append((k, shape, states, need, i, mn + " " + ", ".join(ops), j, m2 + " " + ", ".join(o2)))
                    break
                states += (int(o2[0], 0) + 1) if m2 == "s_nop" and o2 else 1
                if states >= 16:
                    break
    return bad, closest


PK_NEED = 4   # wait states between a packed-f32 op with a mixed low-lane operand selection and the next MFMA (measured: ~50 % wrong at 0, 1e-5 at 1..2, none at >= 3)


def scan_packed(kernels):
    """The hazard that round 4's k_mimi_rowlin and round 2's k_gemm4 actually hit (tools/probes/mfma_hazard/README.md): a VOP3P packed-f32 instruction whose LOW
    result mixes the dword halves of its sources (op_sel:[0,1] -- low = src0.lo (op) src1.HI -- or [1,0]), issued while the matrix pipe is busy and followed at once by
    another MFMA, returns a wrong low result in lanes 48..63 (its last pass).  Measured on MI355X: op_sel:[0,1] forms of v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32
    fail ~50 % of the time with 0 wait states in between, ~1e-5 with 1..2, never with >= 3; plain and [1,1] / [1,0] forms never.  Both mixed forms are flagged.
    hipcc (ROCm 7.2) knows no such hazard: it forms these ops when SLP-vectorising scalar f32 code (a RoPE rotation) and schedules MFMAs right behind them."""
    bad = []
    for k, ins in kernels.items():
        for i, (mn, ops) in enumerate(ins):
            if not (mn.startswith("v_pk_") and mn.endswith("_f32")):
                continue
            m = re.search(r"op_sel:\[([01]),\s*([01])", ",".join(ops))
            if not m or m.group(1) == m.group(2):
                continue
            states = 0
            for j in range(i + 1, min(i + 16, len(ins))):
                m2, o2 = ins[j]
                if m2.startswith(("v_mfma", "v_smfmac")):
                    busy = any(ins[t][0].startswith(("v_mfma", "v_smfmac")) for t in range(max(0, i - 16), i))
                    bad.append((k, states, busy, mn + " " + ", ".join(ops), m2 + " " + ", ".join(o2)))
                    break
                if m2.startswith(("s_cbranch", "s_branch", "s_barrier", "s_endpgm", "s_setpc", "s_swappc")):
                    break
                states += (int(o2[0], 0) + 1) if m2 == "s_nop" and o2 else 1
                if states >= PK_NEED:
                    break
    return bad


def check_library(path):
    kernels = {}
    for co in code_objects(path):
        for k, v in parse(disassemble(co)).items():
            kernels.setdefault(k, []).extend(v)
    return kernels


def main():
    args = sys.argv[1:]
    if not args:
        print(__doc__)
        return 2
    if args[0] == "--asm":
        kernels = parse(open(args[1]).read())
    else:
        kernels = check_library(args[0])
    bad, closest = scan(kernels)
    n_mfma = sum(1 for ins in kernels.values() for mn, _ in ins if mn.startswith("v_mfma"))
    print(f"{len(kernels)} symbols, {n_mfma} MFMA instructions scanned")
    for (k, shape), (st, i, j, m2) in sorted(closest.items(), key=lambda kv: kv[1][0])[:25]:
        print(f"  closest read behind a {shape} MFMA: {st:2d} wait states (need {NEED.get(shape, 12)}) in {k[:90]} by {m2}")
    for b in bad:
        print("VIOLATION: %s: %s read after %d states (need %d)\n    [%d] %s\n    [%d] %s" % b)
    pk = scan_packed(kernels)
    print(f"packed-f32 ops with a mixed low-lane op_sel and an MFMA fewer than {PK_NEED} wait states behind: {len(pk)}")
    for k, st, busy, a, b2 in pk:
        print(f"VIOLATION: {k[:90]}: {st} wait states{'' if busy else ' (no MFMA in the 16 instructions in front: pipe probably idle)'}\n    {a}\n    {b2}")
    return 1 if (bad or pk) else 0


if __name__ == "__main__":
    sys.exit(main())
