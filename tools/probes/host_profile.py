"""cProfile of the host side of one training iteration at a tiny batch (GPU work negligible)."""
import cProfile, os, pstats, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from simplenerf_amd import harness, optim, synth
from simplenerf_amd.data_preprocessors.BatchAssembler01 import BatchAssembler
from simplenerf_amd.loss_functions.LossComputer01 import LossComputer
from simplenerf_amd.models.ModelFactory import get_model
DEV = torch.device('cuda', 0)
cfg = synth.training_configs('f16x3', 32, 32); cfg['sub_batch_size'] = 32
model = get_model(cfg, None).to(DEV).train()
batcher = BatchAssembler(cfg, synth.training_scene(0, 3, 96, 128, 0.05), DEV)
losses = LossComputer(cfg); opt = optim.Adam(list(model.parameters()), lr=5e-4)
it = [20000]
def step():
    it[0] += 1
    return harness.train_one_iter(model, losses, opt, batcher.get_next_batch(it[0]), 32)
for _ in range(5): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(50): step()
torch.cuda.synchronize(); pr.disable()
st = pstats.Stats(pr); st.sort_stats('cumulative').print_stats(38)
