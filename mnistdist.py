#!/usr/bin/env python3
"""Deep MNIST (distributed) -- MI355X-native entrypoint with the reference's flag surface.

    # reference-style launch, one shell per task (reference .idea/MNISTDist.py:13-31):
    python mnistdist.py --job_name=ps     --task_index=0 --ps_hosts=127.0.0.1:2222 --worker_hosts=127.0.0.1:2223,127.0.0.1:2224
    python mnistdist.py --job_name=worker --task_index=0 --ps_hosts=... --worker_hosts=...
    python mnistdist.py --job_name=worker --task_index=1 --ps_hosts=... --worker_hosts=...

    # single GPU / single process:
    python mnistdist.py --synthetic --steps 1000

All ten reference flags keep their names and defaults; ``--steps`` is an alias of
``--training_iter``.  New flags select what the TF script hard-coded or could not do.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from distributed_tensorflow_amd import flags  # noqa: E402

# --- the reference's flags (names, types, defaults: reference M:14-31) ---
flags.DEFINE_string("data_dir", "/tmp/mnist-data", "Directory for string mnist data")
flags.DEFINE_string("ps_hosts", "", "Comma-separated list of hostname:port pairs")
flags.DEFINE_string("worker_hosts", "", "Comma-separated list of hostname:port pairs")
flags.DEFINE_string("job_name", "", "One of 'ps', 'worker'")
flags.DEFINE_integer("task_index", 0, "Index of task within the   job")
flags.DEFINE_integer("hidden_units", 100, "Number of units in the hidden layer of the NN (unused, as in the reference)")
flags.DEFINE_integer("batch_size", 128, "Training batchsize")
flags.DEFINE_integer("training_iter", 10000, "Training iteration")
flags.DEFINE_float("learning_rate", 0.001, "Learning rate")
flags.DEFINE_integer("display_step", 100, "display step")
# --- extensions ---
flags.DEFINE_alias("steps", "training_iter")
flags.DEFINE_string("mode", "sync", "sync (RCCL all-reduce DP) | async (parameter-server semantics)")
flags.DEFINE_string("async_impl", "auto", "auto (push on the hip backend, p2p on the cpu backend) | push (GPU-native: gradients "
                    "and parameters move through peer-to-peer mailbox kernels, csrc/kernels/ps.hip) | p2p (send/recv "
                    "threads; gloo only) | broadcast (reduce + broadcast rounds on collectives)")
flags.DEFINE_integer("staleness", -1, "async: how many of its own steps a worker may run ahead of the parameter server's "
                     "answers (push: 0..8, default 0 = the reference's pull-after-push; broadcast rounds: 0 or 1, default 1)")
flags.DEFINE_string("optimizer", "sgd", "sgd (reference) | adam (TF deep-MNIST tutorial recipe)")
flags.DEFINE_float("keep_prob", 1.0, "dropout keep probability (the reference feeds 1.0)")
flags.DEFINE_string("model", "deep_mnist", "deep_mnist | wide (conv 128/256, fc4096)")
flags.DEFINE_string("width", "", "alias of --model: base | wide")
flags.DEFINE_integer("eval_every", 0, "chief also evaluates the test split every N global steps (0 = only at the end)")
flags.DEFINE_string("backend", "auto", "auto | hip (hand-written MI355X kernels) | cpu (PyTorch oracle, gloo)")
flags.DEFINE_string("logdir", "/tmp/train_logs", "checkpoint directory (reference hard-codes /tmp/train_logs); '' disables")
flags.DEFINE_float("save_model_secs", 600.0, "seconds between chief checkpoints (reference: 600)")
flags.DEFINE_integer("saver_write_version", 2, "checkpoint format the chief writes: 2 = tensor bundle, 1 = TF<0.12 Saver file (both are read)")
flags.DEFINE_boolean("synthetic", False, "use MNIST-shaped synthetic data (automatic when data_dir has no idx files)")
flags.DEFINE_boolean("eval", True, "evaluate on the 10k test split at the end (chief)")
flags.DEFINE_integer("seed", 0, "base random seed")
flags.DEFINE_boolean("hip_graph", True, "allow replaying the single-GPU step as a captured HIP graph (used above 1024 images/step, where it measured faster than eager launches)")
flags.DEFINE_string("dtype", "bf16", "precision of the hip backend: bf16 (bf16 MFMA kernels, fp32 accumulation + fp32 master weights) | fp32 (exact-fp32 MFMA kernel set: the reference's float32 end to end)")
flags.DEFINE_integer("local_rank", -1, "GPU index for this process (default: LOCAL_RANK or rank % device_count)")
flags.DEFINE_float("comm_timeout", 120.0, "collective timeout in seconds (backstop of the failure detection in sync mode)")
flags.DEFINE_float("heartbeat_timeout", 15.0, "a task whose heartbeat on the rendezvous store stops for this long is taken for lost "
                   "(sync mode: the job aborts with a clear message and exit status 3; async mode: only ps tasks are watched)")
flags.DEFINE_integer("health_check_steps", 20, "sync mode on GPUs: every N steps read the job-wide stop flag and the "
                     "peer-to-peer kernels' timeout status (one device sync each)")
flags.DEFINE_string("metrics_file", "", "append JSONL metrics here")
flags.DEFINE_string("comm_backend", "auto", "auto (gloo for the cpu backend, RCCL for GPU tensors) | nccl | gloo "
                    "(gloo also carries GPU tensors, staged through the host: several tasks sharing ONE GPU, which RCCL refuses)")
flags.DEFINE_boolean("p2p_allreduce", False, "sync mode, hip backend: both gradient messages through the peer-to-peer "
                     "exchange fabric (csrc/kernels/dp.hip: sharded dense update, one-shot tail, the step one HIP graph) instead of RCCL")
flags.DEFINE_float("p2p_timeout", 60.0, "bound of every in-kernel wait for a peer (seconds); a wait that runs out poisons the "
                   "worker: no update is applied any more and the job ends with exit status 3")
flags.DEFINE_string("lr_scaling", "none", "none (reference: fixed --learning_rate) | linear | sqrt: scale --learning_rate by "
                    "(global batch / --lr_base_batch) or its square root (utils/lr_schedule.py)")
flags.DEFINE_integer("lr_base_batch", 128, "the batch size --learning_rate is meant for (the reference's default batch)")
flags.DEFINE_integer("lr_warmup_steps", 0, "linear warm-up from --learning_rate to the scaled rate over this many global steps")
flags.DEFINE_string("dense_update", "auto", "sync mode, hip backend over RCCL: auto | sharded (reduce-scatter the fc1 gradient, "
                    "each worker updates 1/N of wd1, all-gather the compute copy) | replicated (all-reduce, every worker "
                    "updates all of it); auto = sharded when there is more than one worker")
flags.DEFINE_string("grad_comm", "fp32", "sync mode, hip backend, bf16 configuration: fp32 (reference arithmetic) | bf16 (opt-in: the "
                    "fc1 weight gradient -- 98 % of the bytes -- is produced, summed across workers and consumed in bf16; "
                    "half the bytes of the dominant collective, not the reference's fp32 gradient sum)")
flags.DEFINE_boolean("trace", False, "roctx ranges around steps/collectives/checkpoints (rocprofv3 --marker-trace)")
flags.DEFINE_integer("fault_inject_step", -1, "test hook: crash this worker after N local steps")
flags.DEFINE_integer("fault_inject_task", -1, "test hook: which worker task crashes (-1 = any)")
flags.DEFINE_integer("fault_mute_step", -1, "test hook (--p2p_allreduce): after N local steps this worker's next step raises none "
                     "of its peer-to-peer flags -- a stalled sender; its peers' waits time out (--p2p_timeout)")
flags.DEFINE_string("dump_params", "", "test hook: write this task's flat fp32 parameter arena to this .npy file at the end")
flags.DEFINE_integer("request_stop_step", -1, "test hook: this worker calls Supervisor.request_stop() after N local steps")
flags.DEFINE_integer("request_stop_task", -1, "test hook: which worker task requests the stop (-1 = any)")

FLAGS = flags.FLAGS


def main(_):
    from distributed_tensorflow_amd import trainer
    trainer.run(FLAGS)
    return 0


if __name__ == "__main__":
    flags.app.run(main)
