"""What do rocprofv3's FETCH_SIZE / WRITE_SIZE count on this chip?  Known byte counts through two regimes:
  stream:   copy 4 GiB -> 4 GiB once (far beyond the 256 MB infinity cache: every byte comes from / goes to HBM), fill 4 GiB, sum 4 GiB
  resident: copy 64 MiB -> 64 MiB and back, 40 times (both buffers fit the infinity cache: after the first pass HBM is not needed)
    python tools/pmc_calibrate.py run            (the workload; run it under rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE --kernel-trace)
    python tools/pmc_calibrate.py report fetch.db write.db out.txt"""
import sys

if sys.argv[1] == "run":
    import torch

    dev = torch.device("cuda", 0)
    big_a = torch.empty(1 << 30, dtype=torch.float32, device=dev).normal_()      # 4 GiB
    big_b = torch.empty_like(big_a)
    torch.cuda.synchronize()
    big_b.copy_(big_a)                     # stream: 4 GiB read + 4 GiB written
    torch.cuda.synchronize()
    big_b.fill_(1.0)                       # stream: 4 GiB written
    torch.cuda.synchronize()
    s = big_a.sum()                        # stream: 4 GiB read
    torch.cuda.synchronize()
    del big_a, big_b
    sa = torch.empty(1 << 24, dtype=torch.float32, device=dev).normal_()         # 64 MiB
    sb = torch.empty_like(sa)
    torch.cuda.synchronize()
    for _ in range(40):                    # resident: 80 copies of 64 MiB
        sb.copy_(sa)
        sa.copy_(sb)
    torch.cuda.synchronize()
    print("done", float(s))
else:
    import sqlite3

    fdb, wdb, out = sys.argv[2:5]
    rows = {}
    for name, db in (("FETCH_SIZE", fdb), ("WRITE_SIZE", wdb)):
        c = sqlite3.connect(db)
        # per dispatch, in order: grid size tells the workload apart (4 GiB vs 64 MiB launches)
        q = "select dispatch_id, kernel_name, sum(value) from counters_collection where counter_name = ? group by dispatch_id order by dispatch_id"
        rows[name] = [(d, k, v) for d, k, v in c.execute(q, (name,))]
    with open(out, "w") as f:
        f.write("# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) --kernel-trace -- python3 tools/pmc_calibrate.py run   (MI355X)\n")
        f.write("# counter values as reported (rocprofv3's unit for both: KB); expected = the bytes the kernel must move, in the same unit\n")
        GiB4, MiB64 = 4 * 1024 * 1024, 64 * 1024   # in KB
        for name in ("FETCH_SIZE", "WRITE_SIZE"):
            r = [x for x in rows[name] if x[2] is not None]
            big = sorted(r, key=lambda x: -x[2])[:3]
            f.write(f"\n{name}: {len(r)} dispatches with the counter\n")
            copies = [x for x in r if "copy" in x[1].lower() or "elementwise" in x[1].lower() or "Memcpy" in x[1]]
            for d, k, v in r[:12]:
                f.write(f"  dispatch {d:4d} {k[:70]:70s} {v:14.0f}\n")
            small = [v for d, k, v in r[-80:]]
            if small:
                f.write(f"  last 80 dispatches (the 64 MiB ping-pong): mean {sum(small) / len(small):.0f}  min {min(small):.0f}  max {max(small):.0f}   (one copy moves {MiB64} KB each way)\n")
            f.write(f"  (a 4 GiB stream is {GiB4} KB)\n")
