"""Per-kernel resources of a hipcc -S --cuda-device-only listing: VGPRs (arch + accumulation), SGPRs, scratch, LDS.
python tools/isa_resources.py file.s [kernel_substring]"""
import re, sys

def main():
    txt = open(sys.argv[1]).read()
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    for m in re.finditer(r"- \.agpr_count:\s+(\d+).*?\.group_segment_fixed_size:\s+(\d+).*?\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+).*?\.sgpr_count:\s+(\d+).*?\.vgpr_count:\s+(\d+)", txt, re.S):
        agpr, lds, name, scratch, sgpr, vgpr = m.groups()
        if want in name:
            print(f"{name:60s} vgpr {vgpr:>3s} (agpr {agpr:>3s}) sgpr {sgpr:>3s} scratch {scratch:>5s} B  lds {lds:>6s} B")

main()
