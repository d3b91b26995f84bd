"""One process per GPU: re-launch a script under torch.distributed.run when it was started as a plain `python script.py --gpus N`.

bench.py's contract lets the driver start N > 1 ranks itself (torch.distributed.run, RANK / WORLD_SIZE in the environment); started WITHOUT a
launcher, `--gpus N` must still mean N ranks on N GPUs -- never a silent one-GPU run that prints an N-GPU line (missing 1).
"""
import os
import socket
import subprocess
import sys


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def under_launcher():
    return "RANK" in os.environ and "WORLD_SIZE" in os.environ


def self_launch(script, argv, n_ranks, require_gpus=True, timeout=None):
    """Run `script argv` as n_ranks ranks of one node (127.0.0.1 rendezvous on a free port) and return the launcher's exit code.  The ranks
    inherit stdout / stderr, so rank 0's JSON line appears on this process's stdout.  Fails loudly if the node has fewer GPUs than ranks."""
    if require_gpus:
        import torch
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < n_ranks:
            raise SystemExit("--gpus %d requested but this node has %d GPU(s): refusing to measure fewer ranks than asked for" % (n_ranks, have))
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC: RCCL's intra-node transports need it on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // n_ranks)))      # each rank its share of the host cores
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % n_ranks, "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), script] + list(argv)
    return subprocess.call(cmd, env=env, timeout=timeout)
