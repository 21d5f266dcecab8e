#!/usr/bin/env python
"""Instruction breakdown of a kernel's layer loop from the gfx950 disassembly (no GPU needed).

    python tools/isa_breakdown.py picaso_amd/csrc/sh.hip 'k_shILi2ELb0ELb0ELb1E' [-o profiles/r05_isa_k_sh4_fast.json]

Compiles the file to device assembly (hipcc -S --cuda-device-only, the library's flags), takes the kernel whose mangled
name contains the pattern, finds its loops (a backward branch to a label) and reports, for the loop with the most
instructions -- the per-layer body -- how many instructions of each kind it holds: fp64 arithmetic by opcode
(v_fma_f64, v_mul_f64, v_add_f64, v_rcp_f64, v_ldexp / v_rndne / v_cvt of the exponentials, min / max of the clips),
compares and selects, moves, memory, scalar, waits.  Only v_*_f64 arithmetic is work the reference's formulas ask for;
the rest (v_mov, v_cndmask, s_waitcnt, address arithmetic) is what an instruction diet can take out.  Static counts of
one pass through the body: branches inside the body (wave-uniform shortcuts) make the dynamic count smaller.
"""
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-fast-math"] + os.environ.get("PICASO_HIPCC_EXTRA", "").split()

KINDS = [
    ("fma_f64", r"^v_fma_f64|^v_fmac_f64|^v_pk_fma_f64"),
    ("mul_f64", r"^v_mul_f64|^v_pk_mul_f64"),
    ("add_f64", r"^v_add_f64|^v_pk_add_f64"),
    ("rcp_rsq_sqrt_f64", r"^v_rcp_f64|^v_rsq_f64|^v_sqrt_f64"),
    ("div_fixup_scale_f64", r"^v_div_"),
    ("exp_parts (ldexp/rndne/cvt/frexp)", r"^v_ldexp_f64|^v_rndne_f64|^v_cvt_|^v_frexp|^v_trunc_f64|^v_floor_f64|^v_fract_f64"),
    ("min_max_f64", r"^v_min_f64|^v_max_f64|^v_min_num_f64|^v_max_num_f64"),
    ("cmp", r"^v_cmp"),
    ("cndmask", r"^v_cndmask"),
    ("mov / readlane / perm", r"^v_mov|^v_accvgpr|^v_readfirstlane|^v_readlane|^v_writelane|^v_perm|^v_swap|^v_pk_mov"),
    ("int / address VALU", r"^v_(add|sub|mul|mad|lshl|lshr|ashr|and|or|xor|not|bfe|bfi|add3|lshl_add|lshl_or|mad_u|mad_i|mul_lo|mul_hi|subrev)_?(co_)?[uib]"),
    ("global / flat / buffer load", r"^(global|flat|buffer)_load"),
    ("global / flat / buffer store", r"^(global|flat|buffer)_store"),
    ("scratch (spill)", r"^scratch_"),
    ("lds", r"^ds_"),
    ("s_waitcnt", r"^s_waitcnt|^s_wait_"),
    ("s_nop / s_delay", r"^s_nop|^s_sleep|^s_delay"),
    ("branch", r"^s_cbranch|^s_branch|^s_setpc|^s_swappc"),
    ("scalar load", r"^s_load|^s_buffer_load"),
    ("scalar ALU", r"^s_"),
]


def disassemble(src):
    out = tempfile.NamedTemporaryFile(suffix=".s", delete=False).name
    subprocess.check_call([HIPCC] + FLAGS + ["-S", "--cuda-device-only", src, "-o", out], stderr=subprocess.DEVNULL)
    with open(out) as fh:
        text = fh.read()
    os.unlink(out)
    return text


def kernel_body(text, pattern):
    names = [m.group(1) for m in re.finditer(r"^(_Z\w+):\s*(?:;.*)?$", text, flags=re.M) if pattern in m.group(1)]
    if not names:
        raise SystemExit("no kernel matching %r" % pattern)
    name = names[0]
    start = text.index("\n" + name + ":")
    end = text.index(".end_amdhsa_kernel", start) if ".end_amdhsa_kernel" in text[start:] else len(text)
    end = text.index("s_endpgm", start)
    # the last s_endpgm of the function: up to the .Lfunc_end label
    fe = re.search(r"^\.Lfunc_end\d+:", text[start:], flags=re.M)
    end = start + fe.start() if fe else end
    meta = text[start:text.index(".end_amdhsa_kernel", start)]
    res = {}
    for key in ("next_free_vgpr", "next_free_sgpr", "accum_offset"):
        m = re.search(r"\.amdhsa_%s\s+(\d+)" % key, meta)
        if m:
            res[key] = int(m.group(1))
    m = re.search(r"; ScratchSize: (\d+)", meta)
    if m:
        res["scratch_bytes"] = int(m.group(1))
    m = re.search(r"; Occupancy: (\d+)", meta)
    if m:
        res["occupancy_waves_per_simd"] = int(m.group(1))
    return name, text[start:end], res


def instructions(body):
    """[(index, label or None, mnemonic, operands)] in program order"""
    out, pending = [], None
    for line in body.split("\n"):
        line = line.split(";")[0].rstrip()
        if not line.strip():
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", line)
        if m:
            pending = m.group(1)
            continue
        s = line.strip()
        if s.startswith(".") or s.endswith(":"):
            continue
        parts = s.split(None, 1)
        out.append((len(out), pending, parts[0], parts[1] if len(parts) > 1 else ""))
        pending = None
    return out


def loops(ins):
    at = {lab: i for i, lab, _, _ in ins if lab}
    found = []
    for i, _, mn, ops in ins:
        if mn.startswith("s_cbranch") or mn == "s_branch":
            tgt = ops.strip()
            if tgt in at and at[tgt] <= i:
                found.append((at[tgt], i))
    return found


def classify(mn):
    for kind, pat in KINDS:
        if re.search(pat, mn):
            return kind
    return "other VALU" if mn.startswith("v_") else "other"


def breakdown(src, pattern):
    text = disassemble(src)
    name, body, res = kernel_body(text, pattern)
    ins = instructions(body)
    lp = loops(ins)
    if not lp:
        raise SystemExit("no loop found in %s" % name)
    lo, hi = max(lp, key=lambda ab: ab[1] - ab[0])
    inner = [ab for ab in lp if ab != (lo, hi) and lo <= ab[0] and ab[1] <= hi]
    counts = collections.Counter(classify(mn) for _, _, mn, _ in ins[lo:hi + 1])
    opc = collections.Counter(mn for _, _, mn, _ in ins[lo:hi + 1])
    n = hi - lo + 1
    fp64 = sum(counts[k] for k in ("fma_f64", "mul_f64", "add_f64", "rcp_rsq_sqrt_f64", "div_fixup_scale_f64",
                                   "exp_parts (ldexp/rndne/cvt/frexp)", "min_max_f64"))
    valu = sum(v for k, v in opc.items() if k.startswith("v_"))
    return {"file": os.path.relpath(src, ROOT), "kernel": name, "registers": res, "instructions_in_kernel": len(ins),
            "layer_loop": {"instructions": n, "valu": valu, "fp64_arithmetic": fp64,
                           "not_arithmetic_valu": valu - fp64, "inner_loops": len(inner),
                           "by_kind": dict(sorted(counts.items(), key=lambda kv: -kv[1])),
                           "top_opcodes": dict(opc.most_common(25))},
            "all_loops": [{"instructions": b - a + 1} for a, b in sorted(lp, key=lambda ab: -(ab[1] - ab[0]))[:6]],
            "flags": " ".join(FLAGS)}


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("-")]
    out = sys.argv[sys.argv.index("-o") + 1] if "-o" in sys.argv else None
    if out:
        args = [a for a in args if a != out]
    rep = breakdown(os.path.join(ROOT, args[0]) if not os.path.isabs(args[0]) else args[0], args[1])
    js = json.dumps(rep, indent=1)
    if out:
        with open(out, "w") as fh:
            fh.write(js + "\n")
    print(js)
