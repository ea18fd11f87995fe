"""One process per GPU: start N ranks of a script on this node (stdlib only -- importing this file never touches torch
or the GPU, so the parent stays a plain launcher; a process that has initialised HIP must not fork/exec workers).

The reference trains on one pinned GPU (talker_trainer.py:704-714) and has no launcher; the contract here is SURVEY 8e:
frames shard over ranks, `torch.distributed` (backend "nccl" = RCCL over xGMI) carries the timing barrier and, in
training, one flat gradient all-reduce per step.  Children get the usual rendezvous environment (RANK, LOCAL_RANK,
WORLD_SIZE, LOCAL_WORLD_SIZE, MASTER_ADDR=127.0.0.1, MASTER_PORT) -- the same one `torch.distributed.run` would set,
so a script launched either way reads the same variables.
"""
import os
import socket
import subprocess
import sys
import time


def free_port():
    s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def rank_env(rank, world, port, base=None):
    env = dict(os.environ if base is None else base)
    env.update({
        "RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_WORLD_SIZE": str(world),
        "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port),
        "HSA_ENABLE_IPC_MODE_LEGACY": "0",  # dmabuf IPC: the only mode the host driver supports (RCCL needs it)
    })
    return env


def _parse_cpulist(text):
    cpus = set()
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def gpu_local_cpus(index, sysfs="/sys/bus/pci/drivers/amdgpu"):
    """CPUs of the NUMA node GPU `index` hangs off (sysfs `local_cpulist` of the index-th amdgpu PCI function in bus order,
    after HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES re-indexing when those are plain integer lists), or None when unknown."""
    try:
        devs = sorted(d for d in os.listdir(sysfs) if d.count(":") == 2)
    except OSError:
        return None
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES"):
        vis = os.environ.get(var)
        if vis:
            try:
                devs = [devs[int(v)] for v in vis.split(",")]
            except (ValueError, IndexError):
                return None
    if not 0 <= index < len(devs):
        return None
    try:
        with open(os.path.join(sysfs, devs[index], "local_cpulist")) as f:
            return _parse_cpulist(f.read()) or None
    except (OSError, ValueError):
        return None


def bind_rank_to_gpu_numa(local_rank):
    """Pin this process to the CPUs next to its GPU (host-side launch latency and the pinned-memory path both cross the socket
    interconnect otherwise).  Only ever narrows the current affinity mask; does nothing when the topology is unknown, when the
    intersection is empty, or when N3DT_NO_AFFINITY=1.  Returns the CPU count bound to, or None."""
    if os.environ.get("N3DT_NO_AFFINITY") == "1" or not hasattr(os, "sched_setaffinity"):
        return None
    local = gpu_local_cpus(local_rank)
    if not local:
        return None
    want = os.sched_getaffinity(0) & local
    if not want:
        return None
    try:
        os.sched_setaffinity(0, want)
    except OSError:
        return None
    return len(want)


def spawn_ranks(argv, world, env=None, timeout=None, poll_s=0.2):
    """Run `sys.executable argv...` as `world` ranks; rank 0 inherits stdout.  Returns the first non-zero exit
    code (the other ranks are then terminated by PID), 0 when every rank succeeded, 124 on timeout."""
    assert world >= 1
    port = free_port()
    procs = []
    for r in range(world):
        procs.append(subprocess.Popen([sys.executable] + list(argv), env=rank_env(r, world, port, env),
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    t0 = time.time()
    rc = 0
    live = set(range(world))
    while live:
        for r in sorted(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            if code != 0 and rc == 0:
                rc = code
                sys.stderr.write("launch: rank %d exited with %d; stopping the other ranks\n" % (r, code))
        if rc != 0 or (timeout is not None and time.time() - t0 > timeout):
            if rc == 0:
                rc = 124
                sys.stderr.write("launch: timeout after %.0f s\n" % (time.time() - t0))
            for r in live:
                procs[r].terminate()
            for r in live:
                try:
                    procs[r].wait(timeout=10)
                except subprocess.TimeoutExpired:
                    procs[r].kill()
            break
        if live:
            time.sleep(poll_s)
    return rc
