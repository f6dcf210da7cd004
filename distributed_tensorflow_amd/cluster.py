"""Cluster description and role/rank resolution.

Reference: ``tf.train.ClusterSpec({"ps": ps_hosts, "worker": worker_hosts})`` +
``tf.train.Server(cluster, job_name, task_index)`` + the ps/worker role switch
(reference .idea/MNISTDist.py:94-107) and ``replica_device_setter`` (M:110-111).

Here a "server" is a rank in one ``torch.distributed`` job (RCCL on GPUs, gloo on
CPU): ps tasks take ranks [0, n_ps), workers [n_ps, n_ps + n_workers).  The
rendezvous TCPStore lives at the first ps host (or the first worker host when
the cluster has no ps), on that host's listed port.  Chief = worker task 0
(reference M:159).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple


def _split_hosts(s) -> List[str]:
    if isinstance(s, (list, tuple)):
        return [h for h in s if h]
    return [h.strip() for h in (s or "").split(",") if h.strip()]


@dataclass
class ClusterSpec:
    jobs: Dict[str, List[str]] = field(default_factory=dict)

    def __init__(self, jobs: Dict[str, object]):
        self.jobs = {name: _split_hosts(hosts) for name, hosts in jobs.items()}

    @staticmethod
    def from_flags(ps_hosts: str, worker_hosts: str) -> "ClusterSpec":
        return ClusterSpec({"ps": ps_hosts, "worker": worker_hosts})

    def num_tasks(self, job: str) -> int:
        return len(self.jobs.get(job, []))

    def job_tasks(self, job: str) -> List[str]:
        return list(self.jobs.get(job, []))

    def task_address(self, job: str, task_index: int) -> str:
        tasks = self.jobs.get(job, [])
        if not 0 <= task_index < len(tasks):
            raise ValueError(f"task_index {task_index} out of range for job {job!r} with {len(tasks)} task(s)")
        return tasks[task_index]

    @property
    def world_size(self) -> int:
        return self.num_tasks("ps") + self.num_tasks("worker")

    def rank_of(self, job: str, task_index: int) -> int:
        self.task_address(job, task_index)  # validates
        if job == "ps":
            return task_index
        if job == "worker":
            return self.num_tasks("ps") + task_index
        raise ValueError(f"job_name must be 'ps' or 'worker', got {job!r}")

    def worker_ranks(self) -> List[int]:
        n = self.num_tasks("ps")
        return list(range(n, n + self.num_tasks("worker")))

    def ps_ranks(self) -> List[int]:
        return list(range(self.num_tasks("ps")))

    def master(self) -> Tuple[str, int]:
        """(host, port) of the rendezvous store: first ps task, else first worker."""
        first = (self.jobs.get("ps") or self.jobs.get("worker") or ["127.0.0.1:29500"])[0]
        host, _, port = first.rpartition(":")
        if not host:
            host, port = first, "29500"
        return host, int(port)


@dataclass
class TaskInfo:
    """What ``tf.train.Server(cluster, job_name, task_index)`` pins down for one process."""

    cluster: ClusterSpec
    job_name: str
    task_index: int

    @property
    def rank(self) -> int:
        return self.cluster.rank_of(self.job_name, self.task_index)

    @property
    def world_size(self) -> int:
        return self.cluster.world_size

    @property
    def is_ps(self) -> bool:
        return self.job_name == "ps"

    @property
    def is_worker(self) -> bool:
        return self.job_name == "worker"

    @property
    def is_chief(self) -> bool:
        return self.is_worker and self.task_index == 0

    @property
    def num_workers(self) -> int:
        return self.cluster.num_tasks("worker")

    @property
    def num_ps(self) -> int:
        return self.cluster.num_tasks("ps")


def replica_device_setter(cluster: ClusterSpec, variable_names: List[str]) -> Dict[str, int]:
    """Variable -> ps task, round-robin in creation order (TF's default placement strategy,
    reference M:110-111).  Kept for parity/introspection: in sync mode parameters are
    replicated; in async mode it decides which ps task owns (applies updates to) each variable."""
    n = max(1, cluster.num_tasks("ps"))
    return {name: i % n for i, name in enumerate(variable_names)}


def single_process_cluster() -> TaskInfo:
    return TaskInfo(ClusterSpec({"ps": [], "worker": ["127.0.0.1:29500"]}), "worker", 0)
