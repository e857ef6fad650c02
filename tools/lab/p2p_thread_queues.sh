#!/bin/bash
# r06: which hardware queue does every stream of the thread-rank peer-to-peer test run on?  One run of tests/test_gpu_dist_native.py's
# p2p_thread_rank_cases under rocprofv3 --kernel-trace (GPU_MAX_HW_QUEUES=32, as the test sets it); the summary counts, per stream, the queue ids
# its kernels were dispatched on, and lists queues that served more than one stream.
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06/p2p_queues; mkdir -p $OUT
export GPU_MAX_HW_QUEUES=32 SMM_HIP_P2P_TIMEOUT_S=5 HSA_ENABLE_IPC_MODE_LEGACY=0
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 500 rocprofv3 --kernel-trace -d $OUT -o trace -- python3 -c "import sys; sys.path.insert(0, '$GRAFT_REPO_ROOT'); sys.path.insert(0, '$GRAFT_REPO_ROOT/tests'); import test_gpu_dist_native as t; t.p2p_thread_rank_cases()" > $OUT/stdout.txt 2> $OUT/stderr.txt )
echo "exit $?"
grep -c "p2p case" $OUT/stderr.txt; grep "libsmm_hip" $OUT/stderr.txt | cut -c1-300
python3 - <<'PY'
import glob, sqlite3, collections, os
db = glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r06/p2p_queues/*.db")[0]
c = sqlite3.connect(db).cursor()
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]; sym = [t for t in tabs if 'kernel_symbol' in t][0]
rows = list(c.execute(f"select d.stream_id, d.queue_id, s.kernel_name, d.start, d.end from {kd} d join {sym} s on d.kernel_id = s.id"))
by_stream = collections.defaultdict(collections.Counter); by_queue = collections.defaultdict(set); p2p_by_queue = collections.defaultdict(set)
for st, q, name, a, b in rows:
    by_stream[st][q] += 1; by_queue[q].add(st)
    if 'p2p' in name: p2p_by_queue[q].add(st)
print("streams", len(by_stream), "queues", len(by_queue), "dispatches", len(rows))
print("streams whose kernels ran on more than one queue:", sum(1 for s, cnt in by_stream.items() if len(cnt) > 1))
shared = {q: sorted(s) for q, s in by_queue.items() if len(s) > 1}
print("queues that served more than one stream:", len(shared))
for q, s in sorted(shared.items()): print("  queue", q, "streams", s)
print("queues on which p2p copy / land / all-reduce kernels of more than one stream ran:", {q: sorted(s) for q, s in p2p_by_queue.items() if len(s) > 1})
PY
