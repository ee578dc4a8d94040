"""per-kernel instruction statistics of a hipcc -S --cuda-device-only listing: MFMAs, accumulator moves, LDS reads, waits, scratch
    python tools/isa_stats.py file.s [name-substring]"""
import re
import sys

s = open(sys.argv[1]).read().split("\n")
pat = sys.argv[2] if len(sys.argv) > 2 else ""
starts = [i for i, l in enumerate(s) if re.match(r"^_Z\S+:\s", l) and pat in l]
for i in starts:
    j = i
    while j < len(s) and "s_endpgm" not in s[j]:
        j += 1
    body = s[i:j]
    txt = "\n".join(body)
    cnt = lambda p: len(re.findall(p, txt))
    print(s[i].split(":")[0][-70:], "| lines", len(body), "mfma", cnt(r"\bv_mfma"), "acc_rd", cnt("v_accvgpr_read"), "acc_wr", cnt("v_accvgpr_write"), "ds_read", cnt(r"\bds_read"),
          "ds_write", cnt(r"\bds_write"), "glds", cnt(r"global_load_lds|\blds$"), "waitcnt", cnt("s_waitcnt"), "barrier", cnt("s_barrier"), "scratch", cnt("scratch_"), "branch", cnt(r"s_cbranch"))
