#!/usr/bin/env python3
"""Static instruction mix of one kernel instance, from the compiler's assembly (no GPU): the tool behind the
vector-issue numbers of DESIGN.md 4c.

    python scripts/asm_mix.py mlf_fused.hip 'k_prep_sweepILi50ELi4'   [top]

Compiles the source with build.py's flags to assembly (device only) and counts the opcodes between the kernel's label and its
s_endpgm; also the number of MFMAs whose C operand is the inline constant 0 and the scratch (spill) instructions."""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ultranest_amd", "csrc"))
import build as B  # noqa: E402


def main():
    src, pat = sys.argv[1], sys.argv[2]
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
    csrc = os.path.join(ROOT, "ultranest_amd", "csrc")
    out = os.path.join(tempfile.gettempdir(), "asm_mix_%s.s" % os.path.splitext(src)[0])
    flags = [f for f in B.FLAGS if f != "-fPIC"] + B.EXTRA_FLAGS.get(src, [])
    subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + flags + ["-S", "--cuda-device-only", "-o", out, os.path.join(csrc, src)],
                   check=True, stderr=subprocess.DEVNULL)
    text = open(out).read()
    m = re.search(r"^(_Z\w*%s\w*):.*?\n(.*?)\n\s*s_endpgm" % re.escape(pat), text, re.S | re.M)
    if not m:
        sys.exit("no kernel matching %r in %s" % (pat, out))
    lines = [ln.strip() for ln in m.group(2).split("\n")]
    lines = [ln for ln in lines if ln and not ln.startswith((";", ".")) and not ln.endswith(":")]
    c = collections.Counter(ln.split()[0] for ln in lines)
    cls = collections.Counter()
    for op, n in c.items():
        k = "mfma" if op.startswith("v_mfma") else "valu" if op.startswith("v_") else "salu" if op.startswith("s_") else \
            "lds" if op.startswith("ds_") else "vmem" if op.startswith(("global_", "buffer_", "scratch_", "flat_")) else "other"
        cls[k] += n
    print(m.group(1))
    print("instructions", len(lines), dict(cls))
    print("mfma with literal-zero C:", sum(1 for ln in lines if ln.startswith("v_mfma") and re.search(r", 0$", ln)))
    print("scratch:", sum(n for op, n in c.items() if op.startswith("scratch_")), " v_readlane/v_writelane:", c["v_readlane_b32"], c["v_writelane_b32"])
    for op, n in c.most_common(top):
        print("  %-32s %d" % (op, n))


if __name__ == "__main__":
    main()
