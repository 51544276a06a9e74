"""Per-kernel identity of the gfx950 device code inside a libbtgpu.so: sha256 (16 hex) of each kernel's disassembly with addresses,
symbol references and branch offsets stripped (the normalisation of scripts/compare_device_code.py).  What it is for: a PMC summary
under profiles/ is evidence for a kernel of ANOTHER build of the library exactly when that kernel's instructions are the same --
bench.py accepts roofline.traffic from such a file only on an equal id (a change elsewhere in the library, e.g. in the exact-payload
kernels, does not un-measure the bank kernel; a change to the bank kernel does).

    python scripts/device_code_ids.py LIB.so                    # print {kernel: id}
    python scripts/device_code_ids.py LIB.so PMC.json [...]     # stamp "kernel_code_sha" into PMC summaries collected on LIB.so
                                                                #   (refused unless sha256(LIB.so) is the file's build_id)
"""
import hashlib, json, os, re, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import compare_device_code as cdc


def short_name(mangled):
    """'_ZN5btgpu14pfb100f_kernelILi256ELb1ELi10ELi255EEEvNS_9PfbParamsE' -> 'pfb100f_kernel<256, true, 10, 255>' (rocprofv3's
    kernel name without return type, namespace and argument list -- the keys of scripts/pmc_hbm_json.py)."""
    n = None
    for tool in ("c++filt", cdc.LL + "/llvm-cxxfilt"):              # (binutils' is in the image; the ROCm LLVM tree ships none)
        try:
            n = subprocess.check_output([tool, mangled], text=True).strip(); break
        except Exception:
            continue
    if not n:
        return mangled
    n = re.sub(r"^void ", "", n).replace("btgpu::", "")
    depth, out = 0, []
    for ch in n:                                   # cut the argument list: the first '(' outside template brackets
        if ch == "<": depth += 1
        elif ch == ">": depth -= 1
        elif ch == "(" and depth == 0: break
        out.append(ch)
    return "".join(out).strip()


def kernel_code_ids(so):
    with tempfile.TemporaryDirectory() as tmp:
        ks = cdc.kernels(cdc.disasm(so, tmp, "a"))
    return {short_name(k): hashlib.sha256("\n".join(v).encode()).hexdigest()[:16] for k, v in ks.items()}


def lookup(ids, pmc_key):
    """The id of the kernel a PMC summary calls `pmc_key` (rocprofv3 truncates long names: a unique prefix match)."""
    if pmc_key in ids:
        return ids[pmc_key]
    m = sorted(k for k in ids if k.startswith(pmc_key))
    if len(m) <= 1:
        return ids[m[0]] if m else None
    # a truncated name that several instantiations share: one id over all of them (equal only if every candidate is unchanged)
    return hashlib.sha256("|".join(k + "=" + ids[k] for k in m).encode()).hexdigest()[:16]


def main():
    so = sys.argv[1]
    ids = kernel_code_ids(so)
    if len(sys.argv) == 2:
        print(json.dumps(ids, indent=1)); return
    bid = hashlib.sha256(open(so, "rb").read()).hexdigest()[:16]
    for path in sys.argv[2:]:
        pj = json.load(open(path))
        if pj.get("build_id") != bid:
            raise SystemExit("%s was collected on build %s, %s is build %s" % (path, pj.get("build_id"), so, bid))
        pj["kernel_code_sha"] = {k: lookup(ids, k) for k in pj["kernels"]}
        pj["kernel_code_sha_note"] = ("sha256/16 of each kernel's normalised gfx950 disassembly in build %s (scripts/device_code_ids.py); "
                                      "bench.py takes this file's bytes for a kernel of another build only on an equal id" % bid)
        json.dump(pj, open(path, "w"), indent=1)
        print(path, pj["kernel_code_sha"])


if __name__ == "__main__":
    main()
