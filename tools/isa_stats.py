"""Register / spill / LDS statistics of the cascade kernels from a device assembly file (hipcc -S --cuda-device-only, or the
.s of -save-temps): python tools/isa_stats.py [--depth] <file.s> [name fragment ...]"""
import re, sys
args = sys.argv[1:]
by_depth = "--depth" in args  # also: where the spill traffic sits, by the loop depth the compiler's block annotations give
args = [a for a in args if a != "--depth"]
s = open(args[0]).read()
frags = args[1:] or ["k_cascade_bulkILi1", "k_passILi1ELi15", "k_cascade_fusedILi1", "k_cascade_fusedILi2"]
md = s[s.find("amdhsa.kernels"):]
for blk in md.split("  - .agpr_count")[1:]:
    nm = re.search(r"\.name:\s+(\S+)", blk).group(1)
    if not any(f in nm for f in frags):
        continue
    g = lambda k: re.search(k + r":\s+(\d+)", blk).group(1)
    body = s[s.find("\n" + nm + ":"):]
    body = body[:body.find(".Lfunc_end")]
    ins = [l.split()[0] for l in body.splitlines() if l.startswith("\t") and not l.strip().startswith((".", ";"))]
    cnt = lambda p: sum(1 for i in ins if i.startswith(p))
    print(f"{nm[:58]:58s} vgpr {g('.vgpr_count'):>3} vspill {g('.vgpr_spill_count'):>2} sgpr-spill {g('.sgpr_spill_count'):>3} lds {g('.group_segment_fixed_size'):>5} "
          f"scratch {g('.private_segment_fixed_size'):>3} | insts {len(ins)} valu {cnt('v_')} salu {cnt('s_')} vmem {cnt('global_') + cnt('flat_')} lds {cnt('ds_')} readlane {cnt('v_readlane')} writelane {cnt('v_writelane')}")
    if by_depth:
        depth, tally = 0, {}
        for l in body.splitlines():
            m = re.search(r"Depth=(\d+)", l)
            if m and ("Loop Header" in l or "in Loop:" in l):
                depth = int(m.group(1))
            elif l.startswith(".LBB") and "Loop" not in l:
                depth = 0
            t = l.split()
            if len(t) and l.startswith("\t"):
                for key in ("v_readlane", "v_writelane", "scratch_load", "scratch_store"):
                    if t[0].startswith(key):
                        tally.setdefault(depth, {}).setdefault(key, 0)
                        tally[depth][key] += 1
        for d in sorted(tally):
            print(f"    loop depth {d}: " + "  ".join(f"{k} {v}" for k, v in sorted(tally[d].items())))
