#!/usr/bin/env python
"""ISA check of the library's rule for data that is rewritten between kernel launches ("scalar-cache hazard").

    python tools/check_scalar_loads.py            # compiles iris_lama_amd/csrc/lama_hip.hip to gfx950 assembly and checks it

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


def assembly():
    out = os.path.join(ROOT, "tools", "_tmp")
    os.makedirs(out, exist_ok=True)
    path = os.path.join(out, "check_scalar_loads.s")
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"),
           "-S", "--cuda-device-only", "-o", path, SRC]
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    return open(path).read().split("\n")


def demangle(names):
    r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True)
    return dict(zip(names, r.stdout.split("\n")))


def scan(lines):
    """-> {kernel: [(line number, instruction, 'kernarg' | 'device')]}"""
    res = collections.OrderedDict()
    kern, kernarg = None, set()
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            kern, kernarg = m.group(1), {(0, 1)}
            res[kern] = []
            continue
        if kern is None:
            continue
        t = l.strip()
        m = re.match(r"s_mov_b64\s+s\[(\d+):(\d+)\],\s*s\[(\d+):(\d+)\]", t)
        if m:
            dst, src = (int(m.group(1)), int(m.group(2))), (int(m.group(3)), int(m.group(4)))
            if src in kernarg:
                kernarg.add(dst)
            else:
                kernarg.discard(dst)
            continue
        m = re.match(r"(s_load_dword\w*|s_buffer_load_dword\w*)\s+(s\d+|s\[\d+:\d+\]),\s*s\[(\d+):(\d+)\]", t)
        if m:
            base = (int(m.group(3)), int(m.group(4)))
            res[kern].append((i + 1, t, "kernarg" if base in kernarg else "device"))
            # the destination overwrites whatever the registers held
            d = re.match(r"s\[(\d+):(\d+)\]", m.group(2))
            lo, hi = (int(d.group(1)), int(d.group(2))) if d else (int(m.group(2)[1:]),) * 2
            for k in list(kernarg):
                if k != (0, 1) and not (k[1] < lo or k[0] > hi):
                    kernarg.discard(k)
            continue
        # any other write to a tracked pair ends its life as a kernarg copy
        m = re.match(r"s_\w+\s+s\[(\d+):(\d+)\],", t) or re.match(r"s_\w+\s+s(\d+)(),", t)
        if m and not t.startswith(("s_cmp", "s_cbranch", "s_waitcnt", "s_bitcmp")):
            lo = int(m.group(1)); hi = int(m.group(2)) if m.group(2) else lo
            for k in list(kernarg):
                if k != (0, 1) and not (k[1] < lo or k[0] > hi):
                    kernarg.discard(k)
    return res


def main():
    lines = assembly()
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
        allow = next((v for pre, v in ALLOWED.items() if name.startswith(pre)), None)
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
