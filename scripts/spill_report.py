"""Register spills per kernel of the built library, from the code objects' own metadata (.vgpr_count / .vgpr_spill_count /
.sgpr_spill_count of every kernel in csrc/build/*.o).  A spill inside a hot loop is silent and expensive: round 6 lost a day's
worth of stage-stamp readings to 25 spilled registers that three extra stamps had caused in k_prep_sweep.
    python scripts/spill_report.py            # every kernel with a vector-register spill"""
import glob
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def kernels():
    """{demangled-ish name: (vgprs, vgpr spills, sgpr spills)} over all objects of the build directory"""
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        for obj in sorted(glob.glob(os.path.join(ROOT, "ultranest_amd", "csrc", "build", "*.o"))):
            dst = os.path.join(tmp, os.path.basename(obj))
            shutil.copy(obj, dst)
            subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", dst], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            for co in glob.glob(dst + ".*gfx950*"):
                notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True).stdout
                name = None
                rec = {}
                for line in notes.splitlines():
                    m = re.match(r"\s*-?\s*\.(name|vgpr_count|vgpr_spill_count|sgpr_spill_count):\s*(\S+)", line)
                    if not m:
                        continue
                    key, val = m.groups()
                    if key == "name" and val.startswith("_Z"):
                        name = val
                        rec = out.setdefault(name, {})
                    elif name and key != "name":
                        rec[key] = int(val)
    return out


if __name__ == "__main__":
    ks = kernels()
    filt = subprocess.run(["c++filt"], input="\n".join(ks), capture_output=True, text=True).stdout.splitlines()
    names = dict(zip(ks, filt))
    bad = {k: v for k, v in ks.items() if v.get("vgpr_spill_count", 0) > 0}
    print("%d kernels, %d with vector-register spills" % (len(ks), len(bad)))
    for k, v in sorted(bad.items(), key=lambda kv: -kv[1]["vgpr_spill_count"]):
        print("%5d spilled  %4d vgprs  %s" % (v["vgpr_spill_count"], v.get("vgpr_count", -1), names[k][:140]))
    sys.exit(0)
