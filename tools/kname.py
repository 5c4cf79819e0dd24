"""Readable kernel names for the profile summaries.  c++filt of this ROCm cannot demangle template arguments of type
__bf16 / _Float16 (Itanium `DF16b` / `DF16_`), so rocprofv3 reports such kernels mangled; this turns
`_ZN4vrag16gemm_bf16_kernelILi3ELi256E...DF16bEEvNS_10GemmParamsE` into `vrag::gemm_bf16_kernel<3, 256, ..., bf16>`."""
import re


def pretty(name: str) -> str:
    m = re.match(r"_ZN4vrag(?:12_GLOBAL__N_1)?(\d+)", name)
    if not m:
        return name
    n = int(m.group(1))
    base = name[m.end(): m.end() + n]
    rest = name[m.end() + n:]
    args = []
    if rest.startswith("I"):
        i = 1
        while i < len(rest) and rest[i] != "E":
            if rest.startswith("Li", i) or rest.startswith("Lb", i):
                j = rest.index("E", i)
                v = rest[i + 2: j]
                args.append(("true" if v == "1" else "false") if rest[i + 1] == "b" else v.replace("n", "-"))
                i = j + 1
            elif rest.startswith("DF16b", i):
                args.append("bf16")
                i += 5
            elif rest.startswith("DF16_", i):
                args.append("f16")
                i += 5
            else:
                return name
    return f"vrag::{base}<{', '.join(args)}>" if args else f"vrag::{base}"


if __name__ == "__main__":
    import sys

    for line in sys.stdin:
        print(pretty(line.strip()))
