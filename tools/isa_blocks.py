"""basic blocks that hold MFMAs in one kernel of a hipcc -save-temps .s file: instruction mix per block.  python tools/isa_blocks.py file.s kernel-substring [dump-block-label]"""
import re
import sys
s = open(sys.argv[1]).read().split("\n")
pat = sys.argv[2]
starts = [i for i, l in enumerate(s) if re.match(r"^_Z\S*:\s*(;.*)?$", l) and pat in l]
if not starts:
    sys.exit("no such kernel; candidates:\n" + "\n".join(l for l in s if re.match(r"^_Z\S*gemm\S*:", l))[:4000])
for st in starts:
    end = next(i for i in range(st, len(s)) if s[i].strip().startswith("s_endpgm"))
    body = s[st:end]
    meta = {}
    for l in s[end:end + 400]:
        m = re.match(r"\s*\.amdhsa_(next_free_vgpr|next_free_sgpr|accum_offset|group_segment_fixed_size|private_segment_fixed_size)\s+(\S+)", l)
        if m:
            meta[m.group(1)] = m.group(2)
    print(s[st][:150], len(body), "lines", meta)
    blk, blocks = "entry", {"entry": []}
    for l in body:
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            blk = m.group(1); blocks[blk] = []
        else:
            blocks[blk].append(l.strip())
    for b, ls in blocks.items():
        ls = [x for x in ls if x and not x.startswith(";") and not x.startswith(".")]
        nm = sum("v_mfma" in x for x in ls)
        if nm or (len(sys.argv) > 3 and b == sys.argv[3]):
            c = lambda f: sum(1 for x in ls if f(x))
            print(f"  {b:12s} {len(ls):5d} insts  mfma {nm:3d}  gload {c(lambda x: 'global_load' in x or 'buffer_load' in x):3d}  ds_read {c(lambda x: x.startswith('ds_read')):3d}  "
                  f"ds_write {c(lambda x: x.startswith('ds_write')):3d}  waitcnt {c(lambda x: 's_waitcnt' in x):3d}  barrier {c(lambda x: 's_barrier' in x):2d}  "
                  f"valu {c(lambda x: x.startswith('v_') and 'mfma' not in x):4d}  salu {c(lambda x: x.startswith('s_') and 'waitcnt' not in x and 'barrier' not in x):4d}  "
                  f"scratch {c(lambda x: 'scratch_' in x):3d}  accmov {c(lambda x: 'v_accvgpr' in x):3d}")
            if len(sys.argv) > 3 and b == sys.argv[3]:
                print("\n".join("      " + x for x in ls))
