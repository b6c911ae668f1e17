"""round 6: BASELINE configs[4] (8 x 3 Gb, d = 0.05) on ONE GPU with more than one genome per block — fewer passes over the
2.4e10 positions against denser block tables (bench.config5_leg(per=..., keys_per_line=...)).
    python tools/config5_blocks.py [genome_mb] "per:kpl" "per:kpl" ...      e.g.  3000 1:0 2:4.5 2:5.5"""
import json
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import bench
    from panagram_amd import engine
    mb = float(sys.argv[1]) if len(sys.argv) > 1 else 3000.0
    dev = torch.device("cuda", 0)
    ctx = engine.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    for spec in sys.argv[2:] or ["1:0"]:
        per, kpl = spec.split(":")
        per, kpl = int(per), float(kpl)
        try:
            out = bench.config5_leg(ctx, dev, types.SimpleNamespace(seed=1234), genome_mb=mb, per=per, keys_per_line=kpl or None)
        except Exception as e:  # noqa: BLE001
            print(f"[per={per} kpl={kpl}] FAILED: {type(e).__name__}: {e}", flush=True)
            ctx.trim()
            torch.cuda.empty_cache()
            continue
        p = out["passes"]
        print(f"[per={per} kpl={kpl or 3}] job {out['value'] / 1e9:.2f} G k-mers/s (with builds {out['value_with_table_builds'] / 1e9:.2f}), "
              f"passes {len(p)}, per pass {out['per_pass_value_mean'] / 1e9:.1f} G, pass ms {[round(x['pass_ms'], 1) for x in p]}, "
              f"probe ms {[round(x['probe_ms'], 1) for x in p]}, extract {round(p[0]['extract_ms'], 1)}, merge {round(p[-1]['merge_ms'], 1)}, "
              f"table keys {p[0]['table_keys']}, GB {p[0]['table_bytes'] / 1e9:.1f}, build s {[round(x['table_build_s'], 2) for x in p]}, m {p[0]['minimizer_length']}, "
              f"rows ok {out['rows_equal_gpu']}, own ok {out['anchors_hold_all_own_kmers']}", flush=True)


if __name__ == "__main__":
    main()
