"""Runs the register-stationary recurrent kernels at the cfg4 shapes a few times for a rocprofv3 --pmc pass:
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_rs_rd -- python3 tools/pmc_lstm_rs.py
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_rs_wr -- python3 tools/pmc_lstm_rs.py
    python3 tools/pmc_lstm_rs.py parse gpurun_out/pmc_rs_rd gpurun_out/pmc_rs_wr"""
import collections, csv, glob, os, re, sys
if len(sys.argv) > 1 and sys.argv[1] == "parse":
    for d, counter, mult in ((sys.argv[2], "FETCH_SIZE", 2.0), (sys.argv[3], "WRITE_SIZE", 1.0)):
        f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
        per = collections.OrderedDict()
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and "lstm_rs" in r["Kernel_Name"]:
                k = (int(r["Dispatch_Id"]), re.sub(r"\(anonymous namespace\)::|^void ", "", r["Kernel_Name"]).split("(")[0])
                per[k] = per.get(k, 0.0) + float(r["Counter_Value"])
        seq = [(k[1], v) for k, v in per.items()]
        # launches in program order: shapes (2080,16), (512,65); per shape 3 x fwd then 3 x bwd
        print(counter, "(KB per launch" + (", doubled per the gfx950 rule)" if mult == 2.0 else ")"))
        for i in range(0, len(seq), 3):
            name = seq[i][0]
            vals = [v * mult / 1e3 for _, v in seq[i:i + 3]]
            print(f"  {name:48s} {sum(vals) / len(vals):9.1f} MB")
    sys.exit(0)
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "speech-separation_amd"))
from dtt_amd._lib import lib, check, ptr, stream_ptr  # noqa: E402
dev, dt, heads, H = "cuda", torch.bfloat16, 2, 512
for (N, L) in ((2080, 16), (512, 65)):
    R = N * L; NT = (N + 31) // 32
    xg = (torch.randn(R, 8 * H, device=dev) * 0.5).to(dt)
    whh = torch.randn(2, 4 * H, H, device=dev) / H ** 0.5
    wt = whh.to(dt).contiguous(); whhT = whh.transpose(1, 2).to(dt).contiguous()
    dy = torch.randn(R, 2 * H, device=dev).to(dt)
    y = torch.empty(R, 2 * H, device=dev, dtype=dt)
    g1 = torch.empty(2, NT, L, 4 * H * 32, device=dev, dtype=dt); c1 = torch.empty(2, NT, L, H * 32, device=dev)
    dG = torch.empty(R, 2, 4 * H, device=dev, dtype=dt); dcb = torch.empty(2, NT, H * 32, device=dev); db = torch.zeros(8 * H, device=dev)
    sync = torch.empty(lib().dtt_lstm_rs_sync_words(N, H), device=dev, dtype=torch.int32)
    for _ in range(3):
        check(lib().dtt_lstm_rs_fwd(ptr(xg), ptr(wt), ptr(y), ptr(g1), ptr(c1), ptr(sync), N, L, heads, H, 1, stream_ptr()), "f")
    for _ in range(3):
        check(lib().dtt_lstm_rs_bwd(ptr(dy), ptr(g1), ptr(c1), ptr(whhT), ptr(dG), ptr(dcb), ptr(db), ptr(sync), N, L, heads, H, 1, stream_ptr()), "b")
    torch.cuda.synchronize()
