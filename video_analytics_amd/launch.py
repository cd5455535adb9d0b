"""One process per GPU, started by the benchmark itself.

The reference's only parallel construct is ``nn.DataParallel`` inside ONE process
(Sheet03/spatialModel.py:133); here every GPU gets its own process (RCCL wants one rank per device)
and the launcher is this module: the parent starts ``world`` fresh children with the
``torch.distributed.run`` environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT),
never touches the GPU itself (a GPU-initialised process must not be re-exec'ed on this pool; the
children are new processes, not exec replacements), forwards the children's stdout/stderr, and
returns non-zero if any rank fails.  An external ``torch.distributed.run`` does the same job: when
WORLD_SIZE is already set nothing is spawned.
"""
import os
import signal
import socket
import subprocess
import sys
import time


NO_GPU_RC = 3  # fewer GPUs visible than ranks asked for (distinct from argparse's 2)


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def needs_spawn(n_ranks, environ=None):
    """True when this process was started bare (no launcher) although more than one rank was asked for."""
    environ = os.environ if environ is None else environ
    return n_ranks > 1 and "WORLD_SIZE" not in environ


def rank_env(rank, world, port, base=None):
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this driver (RCCL needs it)
    return env


def spawn_ranks(argv, world, env=None, timeout=None, poll_s=0.05):
    """Run ``argv`` (a full command line) once per rank and wait.

    Returns the first non-zero exit code (the other ranks are then terminated: a rank that died would
    leave its peers blocked in a collective), 124 on timeout, 0 when every rank succeeded.
    """
    port = free_port()
    procs = [subprocess.Popen(list(argv), env=rank_env(r, world, port, env)) for r in range(world)]
    t0 = time.time()
    rc = 0
    try:
        live = list(procs)
        while live:
            for p in list(live):
                code = p.poll()
                if code is None:
                    continue
                live.remove(p)
                if code != 0 and rc == 0:
                    rc = code if code > 0 else 128 - code
            if rc != 0 or not live:
                break
            if timeout is not None and time.time() - t0 > timeout:
                rc = 124
                break
            time.sleep(poll_s)
    finally:
        for p in procs:  # only reached with live children on failure/timeout/KeyboardInterrupt
            if p.poll() is None:
                p.send_signal(signal.SIGTERM)
        deadline = time.time() + 10
        for p in procs:
            try:
                p.wait(timeout=max(0.1, deadline - time.time()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    return rc


def visible_gpus():
    """Device count without initialising the HIP runtime in this (parent) process."""
    import torch
    try:
        return int(torch.cuda.device_count())
    except Exception:
        return 0


def self_spawn(n_ranks, script, args):
    """bench.py's entry: re-run ``script args`` once per rank.  Returns an exit code."""
    forced = os.environ.get("VA_FORCE_DEVICE") is not None  # rehearsal: all ranks on one GPU over gloo
    have = visible_gpus()
    if have < (1 if forced else n_ranks):
        sys.stderr.write("%s: --gpus %d needs %d visible MI355X GPU(s), found %d; the hot path has no CPU fallback\n"
                         % (os.path.basename(script), n_ranks, 1 if forced else n_ranks, have))
        return NO_GPU_RC
    return spawn_ranks([sys.executable, script] + list(args), n_ranks)
