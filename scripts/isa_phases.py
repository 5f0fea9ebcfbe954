"""Static instructions of k_shade by PHASE, from the compiler's own line table.

    python scripts/isa_phases.py [--kernel k_shade_listILi1ELi0] [--src objective.hip] [-D...]

Compiles csrc/<src> for gfx950 with the Makefile's flags plus -gline-tables-only (-S --cuda-device-only: same code, a `.loc file line`
in front of every instruction group), takes the kernel's listing and attributes every instruction to the source region its `.loc`
names: the regions of shade_body are found by ANCHOR lines in objective.hip (below), the inlined helpers by the function they lie in
(raster_math.h shade_uvz -> barycentrics, common.h wave_segment_reduce9 -> scan, ...).  Small helpers that are called from several
phases (lds_add_f64, ld32, at32, the HIP headers' atomics and math) count towards the phase of the code around them.

k_shade is straight-line code -- four unrolled wave passes, no loop -- so static counts are per-wave dynamic counts EXCEPT for the
branches a wave usually skips; those are listed as their own phases (the deferred-pixel record, taps outside the window, a full vertex
table) so that the common path can be read off.  The scheduler interleaves neighbouring regions: the attribution is exact per
instruction, fuzzy by a few per cent per phase.
"""
import argparse
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "fpc_diffrend_amd", "csrc")
FLAGS = "-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wno-unused-function -Wno-pass-failed".split()

# anchors in objective.hip, in source order: (regex of the FIRST line of the region, phase)
ANCHORS = [
    (r"__device__ __forceinline__ void shade_body\(", "A prologue: id plane + apron into LDS, tables, record slot"),
    (r"auto add_taps = \[&\]", "H texel adds into the LDS window (outside it: memory atomics)"),
    (r"// pass 0 keeps its pixel's taps until the origin is known", "A prologue: id plane + apron into LDS, tables, record slot"),
    (r"auto pixel = \[&\]", "B pixel: id, deferred test"),
    (r"const int t = id - 1;", "C pixel: index + vertex + uv fetch"),
    (r"ShadeKeep K;", "D pixel: barycentrics (rasterize fwd math)"),
    (r"// interpolate \(fit\.py:157\) \+ texture 'linear'", "E pixel: uv interpolation, taps, texels, colour"),
    (r"// background elsewhere \(fit\.py:161\); squared error", "F pixel: loss + d loss/d colour + d colour/d tap position"),
    (r"if \(deferred\) \{      // k_fix reads these back", "R rare: deferred pixel's record (z/w, stores)"),
    (r"float gtu_m = 0\.f, gtv_m = 0\.f;", "M mip only"),
    (r"if \(!MIP && want_tex && nz\) \{", "G pixel: tap cell (x0, y0), hand-over to the window"),
    (r"if \(want_pos\) \{\s*$", "I pixel: gradient chain d uv -> d barycentrics -> nine vertex components"),
    (r"if \(want_pos\)      // \(uniform\)", "J segmented scan of nine components + vertex table adds"),
    (r"if \(MIP\) \{\s*$", "M mip only"),
    (r"auto setup_window = \[&\]", "K window set-up: tap box reductions, barrier, placement"),
    (r"auto flush_window = \[&\]", "L window flush (one float atomic per non-zero cell)"),
    (r"if \(MIP\) \{\s*$", "A prologue: id plane + apron into LDS, tables, record slot"),
    (r"lsum = wave_sum_dpp\(lsum\);", "N epilogue: loss, deferred-bin list, vertex table flush"),
    (r"^// \(r5, measured and dropped: SEVERAL list entries per workgroup", "A prologue: id plane + apron into LDS, tables, record slot"),
]
# functions of the kernels' own headers -> phase (None: counts towards the phase around it)
FUNCS = {
    "common.h": {"wave_segment_reduce9": "J segmented scan of nine components + vertex table adds", "vtable_add": "J segmented scan of nine components + vertex table adds",
                 "vtable_flush": "N epilogue: loss, deferred-bin list, vertex table flush", "vtable_init": "A prologue: id plane + apron into LDS, tables, record slot",
                 "fpcdr_list_item": "A prologue: id plane + apron into LDS, tables, record slot", "fpcdr_decode_bin": "A prologue: id plane + apron into LDS, tables, record slot"},
    "raster_math.h": {"shade_uvz": "D pixel: barycentrics (rasterize fwd math)", "shade_uv_bwd": "I pixel: gradient chain d uv -> d barycentrics -> nine vertex components",
                      "shade_zw": "R rare: deferred pixel's record (z/w, stores)", "shade_pixel": "M mip only", "shade_pixel_bwd": "M mip only"},
    "texsample.h": {"mip_lookup_fwd": "M mip only", "mip_lookup_bwd": "M mip only", "compute_lod": "M mip only", "*": "E pixel: uv interpolation, taps, texels, colour"},
}


def function_ranges(path):
    """{name: [(first line, last line), ...]} of the __device__ functions of a header (brace matching from the definition line)."""
    lines = open(path).read().split("\n")
    out = collections.defaultdict(list)
    i = 0
    while i < len(lines):
        m = re.search(r"__device__[^;(]*?\b(\w+)\s*\(", lines[i])
        if m and not lines[i].strip().startswith("//"):
            name, depth, j, seen = m.group(1), 0, i, False
            while j < len(lines):
                code = re.sub(r"//.*", "", lines[j])
                depth += code.count("{") - code.count("}")
                seen = seen or "{" in code
                if seen and depth <= 0:
                    break
                if not seen and code.rstrip().endswith(";"):
                    break
                j += 1
            if seen:
                out[name].append((i + 1, j + 1))
                i = j
        i += 1
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kernel", default="k_shade_listILi1ELi0")
    ap.add_argument("--src", default="objective.hip")
    ap.add_argument("--listing", default=None, help="an existing -gline-tables-only -S listing instead of compiling")
    args, extra = ap.parse_known_args()
    if args.listing:
        text = open(args.listing).read()
    else:
        with tempfile.TemporaryDirectory() as d:
            out = os.path.join(d, "k.s")
            subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + extra + ["-gline-tables-only", "-S", "--cuda-device-only", args.src, "-o", out],
                                  cwd=CSRC, stderr=subprocess.DEVNULL)
            text = open(out).read()
    lines = text.split("\n")
    files = {}
    for l in lines:
        m = re.match(r'\s*\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"', l) or re.match(r'\s*\.file\s+(\d+)\s+"([^"]+)"', l)
        if m:
            files[int(m.group(1))] = os.path.basename(m.group(2))
    src_lines = open(os.path.join(CSRC, args.src)).read().split("\n")
    # anchors -> sorted (line, phase); each anchor is searched BEHIND the previous one
    marks, at = [], 0
    for rx, phase in ANCHORS:
        hit = next((k for k in range(at, len(src_lines)) if re.search(rx, src_lines[k])), None)
        if hit is None:
            sys.exit(f"anchor not found in {args.src}: {rx}")
        marks.append((hit + 1, phase))
        at = hit + 1
    ranges = {h: function_ranges(os.path.join(CSRC, h)) for h in FUNCS}

    def phase_of(fname, line, current):
        if fname == args.src:
            ph = None
            for ln, p in marks:
                if ln <= line:
                    ph = p
            return ph or current
        if fname in FUNCS:
            for name, spans in ranges[fname].items():
                if any(a <= line <= b for a, b in spans):
                    return FUNCS[fname].get(name, FUNCS[fname].get("*")) or current
            return FUNCS[fname].get("*") or current
        return current      # HIP headers (atomics, math, shuffles): the phase around them

    start = next(i for i, l in enumerate(lines) if re.match(r"^\S*" + re.escape(args.kernel) + r"\S*:", l))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith(".Lfunc_end"))
    cur = marks[0][1]
    table = collections.defaultdict(collections.Counter)
    for l in lines[start + 1:end]:
        t = l.strip()
        m = re.match(r"\.loc\s+(\d+)\s+(\d+)", t)
        if m:
            cur = phase_of(files.get(int(m.group(1)), "?"), int(m.group(2)), cur)
            continue
        if not t or t.startswith((";", ".")) or t.endswith(":") or re.match(r"^\S+:\s", t):
            continue
        op = t.split()[0]
        kind = ("lane-spill" if op in ("v_readlane_b32", "v_writelane_b32") else "valu" if op.startswith("v_") else
                "s_nop" if op == "s_nop" else "salu" if op.startswith("s_") else "lds" if op.startswith("ds_") else
                "vmem" if op.split("_")[0] in ("global", "buffer", "scratch", "flat") else "other")
        table[cur][kind] += 1
        if op.endswith("_dpp") or "_dpp" in op:
            table[cur]["(dpp)"] += 1
        if op.startswith(("v_div", "v_rcp", "v_sqrt", "v_rsq")) or "f64" in op:
            table[cur]["(div/rcp/f64)"] += 1
    kinds = ["valu", "lane-spill", "salu", "s_nop", "lds", "vmem", "(dpp)", "(div/rcp/f64)"]
    tot = collections.Counter()
    print(f"kernel {lines[start].split(':')[0]}   ({args.src}{' ' + ' '.join(extra) if extra else ''})")
    print("%-78s" % "phase" + "".join("%12s" % k for k in kinds) + "   valu %")
    all_valu = sum(c["valu"] + c["lane-spill"] for c in table.values())
    for ph in sorted(table):
        c = table[ph]
        tot.update(c)
        print("%-78s" % ph[:78] + "".join("%12d" % c[k] for k in kinds) + "   %5.1f" % (100.0 * (c["valu"] + c["lane-spill"]) / max(all_valu, 1)))
    print("%-78s" % "total" + "".join("%12d" % tot[k] for k in kinds))
    print("(valu + lane-spill = vector-issue slots; per wave and launch of the kernel on the common path minus the R phase and the memory\n"
          " branch of H; four wave passes = 256 pixels per wave: divide by 4 for a 64-pixel pass)")


if __name__ == "__main__":
    main()
