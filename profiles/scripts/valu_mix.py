"""Static VALU instruction mix of the two dominant raster kernels (default instantiations), from the gfx950 ISA of
raster.o: counts per issue class (classes = the kernels of profiles/microbench/valu_rates.hip whose issue cost the
calibration run measures).  The hot loop exists twice in each kernel (scalar-chase path and LDS-batched path) with the
same body, so the whole-kernel mix is the loop's mix up to the prologue / epilogue (a few % of the instructions).

    python profiles/scripts/valu_mix.py [--json]
"""
import json, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
KERNELS = {"raster_bwd": "raster_bwd_short_kernelILb0ELi1EE", "raster_fwd": "raster_fwd_pk_kernelILb0ELb0ELb0E"}   # (round 6 template parameters: <EXACT, REDUCE> / <EXACT, DEPTH, GROUPS>)

CLASSES = [  # (class, regex on the mnemonic) — first match wins
    ("trans", r"v_(exp|rcp|log|sqrt|rsq)_f32"),
    ("permlane", r"v_permlane"),
    ("dpp", r"_dpp$|v_mov_b32_dpp"),
    ("pk", r"v_pk_"),
    ("cmp", r"v_cmp"),
    ("select", r"v_cndmask|v_min|v_max|v_med3"),
    ("add", r"v_(add|sub|subrev)_f32"),
    ("fma", r"v_(fma|fmac|mul|mad)_f32|v_(fma|fmac|mul)_"),
    ("int_logic", r"v_(and|or|xor|not|lshl|lshr|ashr|bfe|bfi|add_u32|sub_u32|add_co|mul_lo|mul_hi|mad_u|lshlrev|lshrrev|ashrrev|add3|lshl_add|bitop|mbcnt|cvt)"),
    ("mov", r"v_mov|v_readlane|v_readfirstlane|v_writelane|v_accvgpr|v_nop"),
]


def device_asm():
    """gfx950 assembly of raster.hip, built with the flags of csrc/Makefile (as tests/test_isa_properties.py does)"""
    tmp = "/tmp/valu_mix"
    os.makedirs(tmp, exist_ok=True)
    out = os.path.join(tmp, "raster.s")
    csrc = os.path.join(ROOT, "street-gaussians-ns_amd", "csrc")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
                           "-munsafe-fp-atomics", "-fno-slp-vectorize", "-I" + os.path.join(ROOT, "include"), "-I" + csrc,
                           "-S", "--cuda-device-only", "-o", out, os.path.join(csrc, "raster.hip")])
    return open(out).read()


def mix():
    asm = device_asm()
    res = {}
    for name, mangled in KERNELS.items():
        m = re.search(r"^(_Z\w*%s\w*):(.*?s_endpgm)" % re.escape(mangled), asm, re.S | re.M)
        assert m, mangled
        counts, total = {}, 0
        for ln in m.group(2).splitlines():
            parts = ln.split()
            if not parts or not parts[0].startswith("v_"):
                continue
            mn = parts[0]
            # DPP shows as an operand modifier
            key = "dpp" if ("quad_perm" in ln or "row_" in ln) else None
            if key is None:
                for cls, rx in CLASSES:
                    if re.search(rx, mn):
                        key = cls
                        break
            counts[key or "other"] = counts.get(key or "other", 0) + 1
            total += 1
        res[name] = {"symbol": m.group(1)[:80], "valu_static": total, "classes": dict(sorted(counts.items(), key=lambda kv: -kv[1]))}
    return res


if __name__ == "__main__":
    r = mix()
    if "--json" in sys.argv:
        print(json.dumps(r, indent=1))
    else:
        for k, v in r.items():
            print(k, v["valu_static"], {c: round(n / v["valu_static"], 3) for c, n in v["classes"].items()})
