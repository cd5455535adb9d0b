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


KFD_NODES = "/sys/class/kfd/kfd/topology/nodes"


def _kfd_gpu_nodes(root=KFD_NODES, dev_dir="/dev/dri"):
    """GPU agents of the KFD topology in node order: the nodes whose ``properties`` give ``simd_count > 0`` (CPU agents have
    0) and whose render node this process may open (a container that is handed one GPU of eight still sees all eight in
    sysfs, but only its own ``/dev/dri/renderD*``).  Pure file reads: no HIP, no HSA, no driver call."""
    nodes = []
    try:
        names = sorted(os.listdir(root), key=lambda n: int(n) if n.isdigit() else 1 << 30)
    except OSError:
        return nodes
    for name in names:
        props = {}
        try:
            with open(os.path.join(root, name, "properties")) as f:
                for line in f:
                    kv = line.split()
                    if len(kv) == 2:
                        props[kv[0]] = kv[1]
        except OSError:
            continue  # (a node of another container's cgroup reads as permission denied)
        try:
            if int(props.get("simd_count", "0")) <= 0:
                continue
            minor = int(props.get("drm_render_minor", "-1"))
        except ValueError:
            continue
        if dev_dir is not None and minor >= 0:
            node = os.path.join(dev_dir, "renderD%d" % minor)
            if not (os.path.exists(node) and os.access(node, os.R_OK | os.W_OK)):
                continue
        nodes.append({"node": name, "unique_id": props.get("unique_id", "0"), "gfx_target_version": props.get("gfx_target_version")})
    return nodes


def _apply_visible_list(nodes, value):
    """``ROCR_VISIBLE_DEVICES`` / ``HIP_VISIBLE_DEVICES`` semantics: a comma list of indices into ``nodes`` (or, for ROCr,
    ``GPU-<unique id in hex>``); the list ends at the first entry that names no device (how ``-1`` hides every GPU)."""
    out = []
    for tok in value.split(","):
        tok = tok.strip()
        pick = None
        if tok.upper().startswith("GPU-"):
            want = tok[4:].lower().lstrip("0")
            for n in nodes:
                try:
                    if format(int(n["unique_id"]), "x") == want:
                        pick = n
                except ValueError:
                    pass
        else:
            try:
                i = int(tok)
                if 0 <= i < len(nodes):
                    pick = nodes[i]
            except ValueError:
                pass
        if pick is None or pick in out:
            break
        out.append(pick)
    return out


def visible_gpus(environ=None, root=KFD_NODES, dev_dir="/dev/dri"):
    """Number of GPUs a child process will see, WITHOUT touching HIP in this (parent) process: the KFD topology in
    sysfs (``simd_count > 0``, render node accessible) filtered by ``ROCR_VISIBLE_DEVICES`` and then by
    ``HIP_VISIBLE_DEVICES`` / ``CUDA_VISIBLE_DEVICES`` the way the runtimes apply them.  (``torch.cuda.device_count()``
    would call ``hipGetDeviceCount`` here; a process that has initialised the GPU must not start the ranks by exec on
    this pool, and is not provably GPU-free afterwards.)"""
    environ = os.environ if environ is None else environ
    nodes = _kfd_gpu_nodes(root, dev_dir)
    if environ.get("ROCR_VISIBLE_DEVICES") is not None:
        nodes = _apply_visible_list(nodes, environ["ROCR_VISIBLE_DEVICES"])
    for var in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        if environ.get(var) is not None:
            nodes = _apply_visible_list(nodes, environ[var])
            break
    return len(nodes)


def gpu_runtime_loaded():
    """True when this process has mapped the HIP or HSA runtime (what the launcher's parent must never do)."""
    try:
        with open("/proc/self/maps") as f:
            maps = f.read()
    except OSError:
        return False
    return "libamdhip64" in maps or "libhsa-runtime64" in maps


def self_spawn(n_ranks, script, args):
    """bench.py's entry: re-run ``script args`` once per rank.  Returns an exit code."""
    forced = os.environ.get("VA_FORCE_DEVICE") is not None  # rehearsal: all ranks on one GPU over gloo
    have = visible_gpus()
    if have < (1 if forced else n_ranks):
        sys.stderr.write("%s: --gpus %d needs %d visible MI355X GPU(s), found %d; the hot path has no CPU fallback\n"
                         % (os.path.basename(script), n_ranks, 1 if forced else n_ranks, have))
        return NO_GPU_RC
    return spawn_ranks([sys.executable, script] + list(args), n_ranks)
