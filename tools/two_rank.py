"""Diagnosis harness of the two-ranks-on-one-GPU CycleGAN launch (tests/test_steps_gpu.py): runs the launch in its own
process group with a limit, writes stdout / stderr / outcome under gpurun_out/hang/<tag>.*; the group is killed on time-out.
    python tools/two_rank.py <tag> <limit_s> <dump_s> [ENV=VAL ...] [-- extra bench.py args]"""
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from util import run_ranks  # noqa: E402

tag, limit, dump = sys.argv[1], float(sys.argv[2]), sys.argv[3]
rest = sys.argv[4:]
extra = []
if "--" in rest:
    extra = rest[rest.index("--") + 1:]
    rest = rest[:rest.index("--")]
env = dict(os.environ, MIGAN_DP_BACKEND="gloo", MIGAN_DP_SINGLE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0", MIGAN_HANG_DUMP_S=dump)
for kv in rest:
    k, v = kv.split("=", 1)
    env[k] = v
with socket.socket() as s:
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
args = extra or ["--workload", "cyclegan", "--global-batch", "2", "--steps", "2", "--warmup", "1"]
cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
       "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2"] + args + \
      ["--min-seconds", "0", "--no-cpu-baseline", "--no-roofline"]
out_dir = os.path.join(ROOT, "gpurun_out", "hang")
os.makedirs(out_dir, exist_ok=True)
t0 = time.time()
rc, out, err = run_ranks(cmd, ROOT, env, limit)
el = time.time() - t0
with open(os.path.join(out_dir, tag + ".out"), "w") as fh:
    fh.write(out)
with open(os.path.join(out_dir, tag + ".err"), "w") as fh:
    fh.write(err)
line = "%s rc=%s seconds=%.0f env: %s args: %s" % (tag, rc, el, " ".join(rest), " ".join(args))
with open(os.path.join(out_dir, tag + ".rc"), "w") as fh:
    fh.write(line + "\n")
print(line, flush=True)
sys.exit(0 if rc == 0 else 1)
