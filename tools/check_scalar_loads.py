#!/usr/bin/env python
"""ISA check of the library's rule for data that is rewritten between kernel launches ("scalar-cache hazard").

    python tools/check_scalar_loads.py            # compiles iris_lama_amd/csrc/lama_hip.hip to gfx950 assembly and checks it
    python tools/check_scalar_loads.py --wide     # the same for the -DLAMA_WIDE_DM instantiation (liblama_hip_wide.so)

Every scalar memory load (s_load_*, s_buffer_load_*) of every kernel is classified by where its base address comes from:

  * the kernarg segment (s[0:1], or a register pair copied from it): the kernel's own arguments -- always fine;
  * a pointer that was itself loaded from the kernarg segment (a kernel argument / DevParams member), possibly plus an offset
    computed with scalar arithmetic: a scalar load of DEVICE DATA.  Fine for data that only kernels of the SAME stream produce
    (the standard model: a kernel boundary invalidates the scalar cache); NOT wanted for tables the host rewrites between launches
    or that a kernel of another stream produced -- those go through the coherent uniform loads of lama_dev.h (uload_* / pview*).

Because the assembly does not say which buffer a pointer names, the check works with an explicit allow-list: for every kernel, how
many device-data scalar loads it may contain, each entry justified below.  Anything beyond the list -- a new uniform read that the
compiler turned into an s_load -- fails the check and has to be looked at: either it reads same-stream kernel-produced data (add it
here with the reason) or it must use uload_*.  `tests/test_host_logic.py::test_no_scalar_loads_of_host_rewritten_tables` runs this.
"""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "iris_lama_amd", "csrc", "lama_hip.hip")

# kernel (demangled prefix) -> (max number of device-data scalar loads, what they read and why that is safe)
ALLOWED = {
    "lama_dev::k_brushfire_canon": (1, "qsizes of the particle: written by the preceding ray-cast kernel of the same stream (opt-in mode, one stream)"),
    "lama_dev::k_dm_add_obstacles": (1, "counts of the particle: written by kernels of the same stream"),
    "lama_dev::k_occ_max_visited": (1, "counts of the particle: written by kernels of the same stream"),
    "lama_dev::k_ray_replay<": (1, "act_count of the particle: written by k_ray_patches of the same update, lane and stream"),
    "lama_dev::k_raycast": (2, "counts of the particle: written by kernels of the same stream"),
}


def assembly(wide=False):
    out = os.path.join(ROOT, "tools", "_tmp")
    os.makedirs(out, exist_ok=True)
    path = os.path.join(out, "check_scalar_loads_wide.s" if wide else "check_scalar_loads.s")
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"),
           "-S", "--cuda-device-only", "-o", path, SRC] + (["-DLAMA_WIDE_DM"] if wide else [])
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    return open(path).read().split("\n")


def demangle(names):
    r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True)
    return dict(zip(names, r.stdout.split("\n")))


def scan(lines):
    """-> {kernel: [(line number, instruction, 'kernarg' | 'device')]}

    Provenance of scalar registers: a register holds ('k', id, half) when it is half `half` of kernarg-derived pointer number `id`
    (the kernarg segment pointer s[0:1] itself, a copy of it, or kernarg + constant: the hidden arguments); copies by s_mov, the
    add / addc pair that forms kernarg + constant, and spills to VGPR lanes (v_writelane / v_readlane) carry it along; every other
    write to a register clears it.  A scalar load is a kernel-argument load iff both halves of its base pair carry the same id."""
    res = collections.OrderedDict()
    kern, prov, spill, next_id, pending = None, {}, {}, 1, None

    def is_karg(lo, hi):
        a, b = prov.get(lo), prov.get(hi)
        return a is not None and b is not None and a[1] == b[1] and a[2] == 0 and b[2] == 1

    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            kern, prov, spill, next_id, pending = m.group(1), {0: ("k", 0, 0), 1: ("k", 0, 1)}, {}, 1, None
            res[kern] = []
            continue
        if kern is None:
            continue
        t = l.strip()
        if not t or t.startswith((";", ".")):
            continue
        m = re.match(r"(s_load_dword\w*|s_buffer_load_dword\w*)\s+(s\d+|s\[\d+:\d+\]),\s*s\[(\d+):(\d+)\]", t)
        if m:
            res[kern].append((i + 1, t, "kernarg" if is_karg(int(m.group(3)), int(m.group(4))) else "device"))
            d = re.match(r"s\[(\d+):(\d+)\]", m.group(2))
            lo, hi = (int(d.group(1)), int(d.group(2))) if d else (int(m.group(2)[1:]),) * 2
            for r in range(lo, hi + 1):
                prov.pop(r, None)
            continue
        m = re.match(r"s_mov_b64\s+s\[(\d+):(\d+)\],\s*s\[(\d+):(\d+)\]$", t)
        if m:
            d0, s0 = int(m.group(1)), int(m.group(3))
            a, b = prov.get(s0), prov.get(s0 + 1)
            prov.pop(d0, None); prov.pop(d0 + 1, None)
            if a: prov[d0] = a
            if b: prov[d0 + 1] = b
            continue
        m = re.match(r"s_mov_b32\s+s(\d+),\s*s(\d+)$", t)
        if m:
            a = prov.get(int(m.group(2)))
            prov.pop(int(m.group(1)), None)
            if a: prov[int(m.group(1))] = a
            continue
        m = re.match(r"s_add_u32\s+s(\d+),\s*s(\d+),\s*(0x[0-9a-f]+|\d+)$", t)
        if m and prov.get(int(m.group(2)), (0, 0, 1))[2] == 0:
            src = prov[int(m.group(2))]
            prov.pop(int(m.group(1)), None)
            pending = (int(m.group(1)), src[1], next_id)            # low half formed; the high half follows with s_addc_u32
            prov[int(m.group(1))] = ("k", next_id, 0)
            next_id += 1
            continue
        m = re.match(r"s_addc_u32\s+s(\d+),\s*s(\d+),\s*0$", t)
        if m and pending and prov.get(int(m.group(2))) == ("k", pending[1], 1):
            prov.pop(int(m.group(1)), None)
            prov[int(m.group(1))] = ("k", pending[2], 1)
            pending = None
            continue
        m = re.match(r"v_writelane_b32\s+v(\d+),\s*s(\d+),\s*(\d+)$", t)
        if m:
            a = prov.get(int(m.group(2)))
            key = (int(m.group(1)), int(m.group(3)))
            if a: spill[key] = a
            else: spill.pop(key, None)
            continue
        m = re.match(r"v_readlane_b32\s+s(\d+),\s*v(\d+),\s*(\d+)$", t)
        if m:
            prov.pop(int(m.group(1)), None)
            a = spill.get((int(m.group(2)), int(m.group(3))))
            if a: prov[int(m.group(1))] = a
            continue
        # any other instruction that writes scalar registers ends their provenance
        m = re.match(r"(?:s_|v_readfirstlane|v_readlane|v_cmp|v_cmpx)\S*\s+(s\[(\d+):(\d+)\]|s(\d+))\b", t)
        if m and not t.startswith(("s_cmp", "s_cmpk", "s_cbranch", "s_branch", "s_waitcnt", "s_bitcmp", "s_barrier", "s_nop", "s_endpgm", "s_setprio", "s_sleep", "s_dcache")):
            lo = int(m.group(2)) if m.group(2) else int(m.group(4))
            hi = int(m.group(3)) if m.group(3) else lo
            for r in range(lo, hi + 1):
                prov.pop(r, None)
    return res


def main():
    lines = assembly(wide="--wide" in sys.argv)          # --wide: the -DLAMA_WIDE_DM instantiation (liblama_hip_wide.so)
    res = scan(lines)
    names = demangle(list(res))
    bad = 0
    total_dev = 0
    for k, loads in res.items():
        dev = [x for x in loads if x[2] == "device"]
        if not dev:
            continue
        total_dev += len(dev)
        name = names[k]
        name = name[5:] if name.startswith("void ") else name
        allow = next((v for pre, v in ALLOWED.items() if name.replace("lama_dev_wide::", "lama_dev::").startswith(pre)), None)
        limit = allow[0] if allow else 0
        status = "ok" if len(dev) <= limit else "FAIL"
        print(f"{status:4s} {len(dev)} device-data scalar load(s), {limit} allowed: {name[:110]}")
        if len(dev) > limit:
            bad += 1
            for ln, ins, _ in dev:
                print(f"        line {ln}: {ins}")
        elif allow:
            print(f"        ({allow[1]})")
    print(f"{len(res)} kernels / device functions, {total_dev} scalar loads of device data, {bad} kernel(s) over their allowance")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
