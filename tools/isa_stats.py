"""Register / spill / LDS statistics of the cascade kernels from a device assembly file (hipcc -S --cuda-device-only, or the
.s of -save-temps): python tools/isa_stats.py <file.s> [name fragment ...]"""
import re, sys
s = open(sys.argv[1]).read()
frags = sys.argv[2:] or ["k_cascade_bulkILi1", "k_passILi1ELi15", "k_cascade_fusedILi1", "k_cascade_fusedILi2"]
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
