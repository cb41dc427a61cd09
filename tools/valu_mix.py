#!/usr/bin/env python3
"""Developer tool: static VALU-vs-MFMA instruction mix of the tile loop of every MFMA stack kernel
(fp32 MFMA and other vector instructions do not co-issue on gfx950, so non-MFMA VALU per MFMA is the
overhead factor of a kernel).   python tools/valu_mix.py [file.hip ...]"""
import collections, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pwclonet_pylidarslam_amd", "csrc")
files = sys.argv[1:] or ["fused_hoisted.hip", "fused_layers.hip", "fused_sa.hip"]
for f in files:
    with tempfile.NamedTemporaryFile(suffix=".s") as tmp:
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
                        "-fno-slp-vectorize", "-S", "--cuda-device-only", os.path.join(CSRC, f), "-o", tmp.name],
                       check=True, stderr=subprocess.DEVNULL)
        text = open(tmp.name).read()
    for m in re.finditer(r"^(_ZN5pwclo\w+):[^\n]*\n(.*?)s_endpgm", text, re.S | re.M):
        name, body = m.group(1), m.group(2)
        lines = body.split("\n")
        heads = [i for i, l in enumerate(lines) if "Inner Loop Header" in l]
        if not heads:
            continue
        loop = lines[heads[-1]:]
        ops = collections.Counter(l.split()[0] for l in loop if re.match(r"\s+v_", l))
        mfma = sum(v for k, v in ops.items() if k.startswith("v_mfma"))
        if mfma < 8:
            continue
        other = sum(ops.values()) - mfma
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        dem = dem.replace("pwclo::", "").split("(")[0].replace("void ", "")
        top = ", ".join("%s %d" % kv for kv in ops.most_common(6) if not kv[0].startswith("v_mfma"))
        print("%-46s mfma %5d  valu %5d  (%.2f / mfma)  %s" % (dem[:46], mfma, other, other / mfma, top))
