"""VGPRs / spills / occupancy per kernel from hipcc -Rpass-analysis=kernel-resource-usage output (stderr saved to a file)."""
import re, sys
t = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
blocks = re.split(r'remark: Function Name: ', t)[1:]
for b in blocks:
    name = b.split()[0]
    if pat not in name: continue
    v = re.search(r' VGPRs: (\d+)', b).group(1); sp = re.search(r'VGPRs Spill: (\d+)', b).group(1); oc = re.search(r'Occupancy \[waves/SIMD\]: (\d+)', b).group(1)
    print(f"{name[:90]:90s} VGPR {v:>4s} spill {sp:>3s} occ {oc}")
