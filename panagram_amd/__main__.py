"""CLI surface kept from the reference (`panagram index <samples.tsv> -k K [-o prefix] [-c cores]
[--prepare]`, panagram/__main__.py:154-194, index.py:90-123) plus the process-level seam of
cpp/run_anchor (`run_anchor <ngenomes> <root> [<name> <fasta>]...`)."""
import argparse
import os
import sys


def main(argv=None):
    ap = argparse.ArgumentParser(prog="panagram_amd")
    sub = ap.add_subparsers(dest="cmd", required=True)
    ix = sub.add_parser("index", help="Anchor k-mer bitvectors to reference FASTA files to create pan-kmer bitmap")
    ix.add_argument("input", metavar="config_file")
    ix.add_argument("-o", "--prefix", default=None)
    ix.add_argument("-k", type=int, default=21)
    ix.add_argument("-c", "--cores", type=int, default=1)
    ix.add_argument("-p", "--prepare", action="store_true")
    ix.add_argument("--anchor_genomes", nargs="*", default=None)
    # one process per GPU under torchrun: LOCAL_RANK picks the GPU, RANK / WORLD_SIZE the anchor genomes
    ix.add_argument("--device", type=int, default=int(os.environ.get("LOCAL_RANK", "0")))
    ix.add_argument("--export_kmc", action="store_true", help="also write kmc/bitvec{i} (KMC1 layout)")
    ix.add_argument("--kmc.use_existing", dest="use_existing", action="store_true")
    ra = sub.add_parser("run_anchor", help="argv-compatible with the reference's cpp/run_anchor")
    ra.add_argument("args", nargs="+")
    ra.add_argument("--device", type=int, default=0)
    a = ap.parse_args(argv)
    if a.cmd == "index":
        from .index import KMC, Index
        idx = Index(a.input, prefix=a.prefix, k=a.k, cores=a.cores, prepare=a.prepare,
                    anchor_genomes=a.anchor_genomes, device=a.device, export_kmc=a.export_kmc,
                    kmc=KMC(use_existing=a.use_existing))
        idx.run()
        return 0
    from .index import run_anchor_cli
    return run_anchor_cli(a.args, a.device)


if __name__ == "__main__":
    sys.exit(main())
