#!/bin/bash
# LDS bank-conflict / activity counters of the conv kernels: tools/pmc_lds.sh <lib.so> <kbench filters...>   (rocprofv3 --pmc only, no tracing)
cd $GRAFT_REPO_ROOT
export CVAE_HIP_LIB=$GRAFT_REPO_ROOT/causal_vae_amd/$1; shift
out=gpurun_out/pmc_lds
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS -d $GRAFT_REPO_ROOT/$out/p1 -o p1 --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/kbench.py --iters 3 "$@" > $GRAFT_REPO_ROOT/$out/log1.txt 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES -d $GRAFT_REPO_ROOT/$out/p2 -o p2 --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/kbench.py --iters 3 "$@" > $GRAFT_REPO_ROOT/$out/log2.txt 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob, collections
for p in ("p1", "p2"):
    for f in glob.glob(f"gpurun_out/pmc_lds/{p}/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:70]
            if "conv" not in k: continue
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
        for k, d in agg.items():
            print(p, k)
            for c, v in d.items(): print(f"      {c:34s} {v / cnt[(k, c)]:14.0f} per launch")
PY
