"""GPU box: why does the bench line's `cli` block (1 GB) take as long as 5 GB does from a shell?  Runs the binary on 1M x 1 kb
(a) from a bare python process, (b) from a process that has initialised the GPU and holds 45 GB on it (what bench.py is when it
reaches the block), (c) as (b) with the stdout / stderr pipes of subprocess.run(capture_output=True).  CIRCKIT_CLI_TIMING=1."""
import os, subprocess, sys, time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
exe = os.path.join(ROOT, "circkit_amd", "circkit")
R, L = 1_000_000, 1000
src = "/dev/shm/probe_in.fasta"
rng = np.random.default_rng(1)
rows = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(R, L))]
out = np.empty((R, L + 11), dtype=np.uint8)
out[:, 0] = ord(">"); out[:, 1] = ord("r"); out[:, 9] = 10; out[:, L + 10] = 10
idx = np.arange(R)
for d in range(7):
    out[:, 8 - d] = 48 + (idx // 10 ** d) % 10
out[:, 10:L + 10] = rows
out.tofile(src)
del out, rows


def run(label, capture):
    env = dict(os.environ, CIRCKIT_CLI_TIMING="1")
    for sink in ("/dev/null", "/dev/shm/probe_out.fasta"):
        t0 = time.perf_counter()
        r = subprocess.run([exe, "canonicalize", src, "-o", sink], env=env, capture_output=capture,
                           stderr=None if capture else subprocess.PIPE)
        dt = time.perf_counter() - t0
        err = r.stderr.decode(errors="replace")
        print("%-44s %-26s %.3f s wall" % (label, sink, dt))
        for line in err.splitlines():
            if "busy" in line or "main()" in line:
                print("      " + line.strip())
        sys.stdout.flush()


run("bare python", False)
run("bare python", False)
import torch
torch.cuda.init()
x = torch.empty(45 << 30, dtype=torch.uint8, device="cuda:0")
x.fill_(1)
torch.cuda.synchronize()
run("GPU initialised, 45 GB held", False)
run("GPU initialised, 45 GB held, capture_output", True)
del x
torch.cuda.empty_cache()
run("GPU initialised, memory released", False)
for f in (src, "/dev/shm/probe_out.fasta"):
    try:
        os.unlink(f)
    except OSError:
        pass
