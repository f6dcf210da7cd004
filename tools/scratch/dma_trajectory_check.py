"""End-to-end race screen of the LDS-DMA GEMMs: the kernels are bit-identical to the register-staged ones, so a whole
training trajectory must be too.  Prints a digest of the parameters after N steps; run once per DTF_GEMM_DMA mask and
compare (tools/scratch/dma_trajectory_check.sh)."""
import hashlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from distributed_tensorflow_amd.engine import DeviceDataset, HipEngine
from distributed_tensorflow_amd.models.spec import ModelSpec
steps, B, opt = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
ds = DeviceDataset.synthetic(60000, seed=1234)
e = HipEngine(ModelSpec(), B, ds, optimizer=opt, learning_rate=1e-3, seed=3, keep_prob=0.75, fuse_dense_update=(opt == "sgd"))
e.init_params(seed=1)
for _ in range(steps):
    e.train_step()
torch.cuda.synchronize()
h = hashlib.sha256(e.params.cpu().numpy().tobytes()).hexdigest()[:16]
print(f"mask {os.environ.get('DTF_GEMM_DMA', 'default')} steps {steps} B {B} {opt}: params {h} loss {e.last_metrics()[0]:.6f}", flush=True)
