cd ${GRAFT_REPO_ROOT:-/root/repo}
export PG_SHAPES="--d 0.01;--d 0.03;--d 0.05;--d 0.1;--genomes 27 --genome-mb 40 --d 0.05"
echo "## default"; bash tools/lines.sh def
echo "## minimizer 15 / 17"; PG_EACH="--minimizer 15;--minimizer 17" bash tools/lines.sh m
echo "## 256-byte lines (PG_TABLE_SLOTS=16)"; PG_TABLE_SLOTS=16 bash tools/lines.sh s16
echo "## keys per line 1.5 / 4"; PG_EACH="--keys-per-bucket 1.5;--keys-per-bucket 4" bash tools/lines.sh kpb
echo "## one launch per genome"; PG_EXTRA="--per-genome-launches" bash tools/lines.sh pg
