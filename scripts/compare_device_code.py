"""Kernel-by-kernel comparison of the gfx950 code objects of two builds of libbtgpu.so (addresses, symbol references and branch
offsets stripped): which kernels are instruction-identical, which differ, which exist on one side only.
    python scripts/compare_device_code.py old.so new.so
(make -C gr-bluetooth_amd/csrc verify-device-code compares a rebuild of the SAME sources byte for byte; this one is for two builds
whose kernel sets differ -- e.g. the build the whole GPU suite ran on against the committed one.)"""
import os, re, subprocess, sys, tempfile
LL = "/opt/rocm/lib/llvm/bin"


def disasm(so, tmp, tag):
    fat, co = os.path.join(tmp, tag + ".fat"), os.path.join(tmp, tag + ".co")
    # (an output file is named on purpose: without one llvm-objcopy rewrites its INPUT in place -- the library's bytes, and with
    # them the build id bench.py stamps, would change)
    subprocess.check_call([LL + "/llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, so, os.path.join(tmp, tag + ".copy")])
    subprocess.check_call([LL + "/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + fat,
                           "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
    return subprocess.check_output([LL + "/llvm-objdump", "-d", "--no-show-raw-insn", co], text=True)


def kernels(text):
    d, cur = {}, None
    for ln in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.*)>:", ln)
        if m:
            cur = m.group(1); d[cur] = []
        elif cur is not None:
            t = re.sub(r"//.*$", "", ln)
            t = re.sub(r"^\s*[0-9a-f]+:", "", t)
            t = re.sub(r"<[^>]*>", "", t).strip()
            t = re.sub(r"^(s_cbranch\w*|s_branch|s_call_b64\s+\S+,)\s+\d+", r"\1 T", t)
            if t:
                d[cur].append(t)
    return d


def main():
    with tempfile.TemporaryDirectory() as tmp:
        a, b = kernels(disasm(sys.argv[1], tmp, "a")), kernels(disasm(sys.argv[2], tmp, "b"))
    def demangle(n):
        try:
            return subprocess.check_output([LL + "/llvm-cxxfilt", n], text=True).strip().split("(")[0].replace("void ", "")
        except Exception:
            return n
    same = [k for k in a if k in b and a[k] == b[k]]
    diff = [k for k in a if k in b and a[k] != b[k]]
    # a template that gained a (defaulted) parameter has new mangled names: pair what is left by content
    left_a, left_b = [k for k in a if k not in b], [k for k in b if k not in a]
    renamed = []
    for k in list(left_a):
        for j in left_b:
            if a[k] == b[j]:
                renamed.append((k, j)); left_a.remove(k); left_b.remove(j)
                break
    print("%d / %d kernels: %d instruction-identical under the same name, %d under a changed name, %d differ"
          % (len(a), len(b), len(same), len(renamed), len(diff)))
    for k in diff:
        print("  differs:", demangle(k))
    for k, j in renamed:
        print("  identical, renamed:", demangle(k), "->", demangle(j))
    for k in left_a:
        print("  only in the first :", demangle(k))
    for k in left_b:
        print("  only in the second:", demangle(k))


if __name__ == "__main__":
    main()
