"""One process per GPU, started by the benchmark itself (BASELINE.json configs[3]).

``python bench.py --gpus N`` must work without ``torch.distributed.run``: :func:`spawn_ranks` starts N
fresh child interpreters of the same command line with the torchrun environment (``RANK``,
``LOCAL_RANK``, ``WORLD_SIZE``, ``LOCAL_WORLD_SIZE``, ``MASTER_ADDR`` = 127.0.0.1, ``MASTER_PORT``)
*before the parent has made any GPU call* -- the parent never touches the GPU, it only forwards rank
0's stdout (the ONE JSON line), sends the other ranks' stdout to its stderr, and returns non-zero if
any child does (the remaining children are then terminated by PID).  The reference starts its ranks
with ``torch.distributed.launch`` and reads the same variables (``opencood/tools/train_ddp.py:62-76``,
``opencood/utils/multi_gpu_utils.py:16-38``); a process started by torchrun already carries them and
is never re-launched (:func:`needs_launch`).

``share_device`` (``bench.py --share-device``) is the rehearsal mode for a 1-GPU box: every rank
uses ``cuda:0`` and the process group is gloo, because RCCL refuses two ranks on one device
("Duplicate GPU detected").  It exercises the launcher, the environment, the DDP wrapper and the
aggregation, not xGMI.
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys
import threading
import time
from typing import Dict, List, Optional, Sequence, Tuple

ENV_LAUNCHED = "GENCOMM_LAUNCHED"          # set in every child: "this process is a rank, do not launch again"
ENV_SHARE_DEVICE = "GENCOMM_SHARE_DEVICE"  # every rank on cuda:0, gloo process group


def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def needs_launch(gpus: int, environ=None) -> bool:
    """True when ``--gpus N > 1`` was asked of a process that is not already one of N ranks."""
    env = os.environ if environ is None else environ
    return gpus > 1 and "RANK" not in env and "WORLD_SIZE" not in env and ENV_LAUNCHED not in env


def rank_env(rank: int, world: int, port: int, share_device: bool = False, base=None) -> Dict[str, str]:
    env = dict(os.environ if base is None else base)
    env.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_WORLD_SIZE": str(world),
                "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), ENV_LAUNCHED: "1",
                "HSA_ENABLE_IPC_MODE_LEGACY": env.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")})   # dmabuf IPC (RCCL needs it on this pool)
    if share_device:
        env[ENV_SHARE_DEVICE] = "1"
    return env


def device_index(local_rank: int, environ=None) -> int:
    """The HIP device a rank uses: its LOCAL_RANK, or 0 for every rank in the share-device rehearsal."""
    env = os.environ if environ is None else environ
    return 0 if env.get(ENV_SHARE_DEVICE) == "1" else local_rank


def backend(environ=None) -> str:
    """"nccl" (= RCCL on ROCm) unless the ranks share one device."""
    env = os.environ if environ is None else environ
    return "gloo" if env.get(ENV_SHARE_DEVICE) == "1" else "nccl"


def _pump(stream, sinks: List, keep: Optional[List[str]]):
    for line in iter(stream.readline, ""):
        if keep is not None:
            keep.append(line)
        for s in sinks:
            s.write(line)
            s.flush()
    stream.close()


def spawn_ranks(argv: Sequence[str], nproc: int, *, share_device: bool = False, port: Optional[int] = None,
                timeout: Optional[float] = None, echo: bool = True, env=None) -> Tuple[int, str]:
    """Start ``nproc`` children of ``argv`` (a full command line, interpreter first), one per rank.

    Returns ``(returncode, rank-0 stdout)``.  ``returncode`` is 0 only if every child exited 0; the first
    non-zero exit (or the timeout) terminates the others and is returned (124 for the timeout).
    """
    if nproc < 1:
        raise ValueError("nproc must be >= 1")
    port = free_port() if port is None else port
    procs: List[subprocess.Popen] = []
    threads: List[threading.Thread] = []
    rank0_out: List[str] = []
    for r in range(nproc):
        try:
            p = subprocess.Popen(list(argv), env=rank_env(r, nproc, port, share_device, env), stdout=subprocess.PIPE, text=True, bufsize=1)
        except Exception:
            for q in procs:          # a rank that cannot be started must not leave the earlier ones waiting at the rendezvous
                q.terminate()
            raise
        procs.append(p)
        sinks = ([sys.stdout] if echo else []) if r == 0 else [sys.stderr]
        t = threading.Thread(target=_pump, args=(p.stdout, sinks, rank0_out if r == 0 else None), daemon=True)
        t.start()
        threads.append(t)
    deadline = None if timeout is None else time.monotonic() + timeout
    rc = 0
    live = set(range(nproc))
    while live and rc == 0:
        for r in sorted(live):
            code = procs[r].poll()
            if code is not None:
                live.discard(r)
                if code != 0:
                    rc = code if code > 0 else 128 - code
                    print(f"[launch] rank {r} exited with {code}; stopping {len(live)} other rank(s)", file=sys.stderr)
                    break
        if deadline is not None and time.monotonic() > deadline and live:
            rc = 124
            print(f"[launch] timeout after {timeout} s; stopping {len(live)} rank(s)", file=sys.stderr)
        if live and rc == 0:
            time.sleep(0.05)
    for r in live:                      # exact PIDs of our own children only
        procs[r].terminate()
    for r in live:
        try:
            procs[r].wait(timeout=15)
        except subprocess.TimeoutExpired:
            procs[r].kill()
            procs[r].wait()
    for t in threads:
        t.join(timeout=5)
    return rc, "".join(rank0_out)


def self_launch(script: str, args: Sequence[str], nproc: int, *, share_device: bool = False, timeout: Optional[float] = None) -> int:
    """Re-run ``python script args`` as ``nproc`` ranks; the exit code for the parent.  Call before any GPU call."""
    rc, _ = spawn_ranks([sys.executable, script, *args], nproc, share_device=share_device, timeout=timeout)
    return rc
