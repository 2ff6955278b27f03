"""Development tool: register / spill / LDS / scratch figures of every kernel and device function in a saved .s file
(-save-temps).  usage: kernel_resources.py file.s [name filter]"""
import re, sys, subprocess
txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
def demangle(n):
    try: return subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", n], capture_output=True, text=True).stdout.strip()
    except Exception: return n
# kernels: amdhsa metadata
for m in re.finditer(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.sgpr_count:\s+(\d+)\n\s+\.sgpr_spill_count:\s+(\d+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)", txt):
    pass
blocks = re.split(r"\n  - \.agpr_count:", txt)
for b in blocks[1:]:
    g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, b) or [None, "?"])[1]
    name = demangle(g("name"))
    if flt and flt not in name: continue
    print("%-110s sgpr %s spill %s | vgpr %s spill %s | scratch %s | lds %s" % (name[:110], g("sgpr_count"), g("sgpr_spill_count"), g("vgpr_count"), g("vgpr_spill_count"), g("private_segment_fixed_size"), g("group_segment_fixed_size")))
# device functions: the ; NumVgprs comments
for m in re.finditer(r"; -- End function\n(?:.*\n){0,3}?\s*\.section.*\n|^(\S+):\s*; @(\S+)\n", txt, re.M):
    pass
for m in re.finditer(r"\.type\s+(\S+),@function\n(?:.*\n)*?; NumSgprs: (\d+)\n; NumVgprs: (\d+)\n(?:.*\n)*?; ScratchSize: (\d+)", txt):
    name = demangle(m.group(1))
    if "chain_runner" in name and (not flt or flt in name or True):
        print("FUNC %-100s sgpr %s vgpr %s scratch %s" % (name[:100], m.group(2), m.group(3), m.group(4)))
