import os, sys, types, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench
dev = torch.device("cuda", 0)
args = types.SimpleNamespace(d=0.01, seed=1234)
for roomy in ("1", "0", "1", "1"):
    os.environ["PG_TABLE_ROOMY"] = roomy
    o = bench.e2e_leg(dev, args, 8, [20_000_000] * 5, 21)
    print("roomy", roomy, {k: round(o[k], 3) for k in ("seconds", "read_parse_sketch_s", "table_insert_s", "anchor_and_write_s")}, flush=True)
