cd $GRAFT_REPO_ROOT
one() { python bench.py "$@" 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print(d['ms_per_step'])"; }
for p in n h l n h; do echo "prio=$p cfg3/8 launched $(NIW_SIDE_PRIORITY=$p one --config cfg3 --shard-of 8 --lean --steps 200 --hip-graph off)  cfg2/8 $(NIW_SIDE_PRIORITY=$p one --config cfg2 --shard-of 8 --lean --steps 100 --hip-graph off) cfg3 $(NIW_SIDE_PRIORITY=$p one --config cfg3 --lean --steps 50 --hip-graph off)"; done
