"""Diagnostic: register / spill / scratch figures of every solver kernel variant from the device assembly
(hipcc -S --cuda-device-only ... -o file.s)."""
import re, sys
txt = open(sys.argv[1]).read()
for b in txt.split('  - .agpr_count:')[1:]:
    name = re.search(r'\.name:\s+(\S+)', b).group(1)
    if 'map_score' not in name:
        continue
    g = lambda k: re.search(r'\.%s:\s+(\d+)' % k, b).group(1)
    short = name.replace('_ZN4muse16map_score_kernelINS_', '').replace('EvNS_9BatchArgsE', '')
    print(f"{short:70s} vgpr {g('vgpr_count'):>3s} agpr {b.split(chr(10))[0].strip():>3s} vspill {g('vgpr_spill_count'):>3s} sspill {g('sgpr_spill_count'):>3s} scratch {g('private_segment_fixed_size')}")
