import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from simplenerf_amd import harness, optim, synth
from simplenerf_amd.data_preprocessors.BatchAssembler01 import BatchAssembler
from simplenerf_amd.loss_functions.LossComputer01 import LossComputer
from simplenerf_amd.models.ModelFactory import get_model
DEV = torch.device('cuda', 0)
for rays, sparse, sub in ((32, 32, 32), (256, 256, 256), (2048, 2048, 2048)):
    cfg = synth.training_configs('f16x3', rays, sparse); cfg['sub_batch_size'] = sub
    model = get_model(cfg, None).to(DEV).train()
    batcher = BatchAssembler(cfg, synth.training_scene(0, 3, 96, 128, 0.05), DEV)
    losses = LossComputer(cfg); opt = optim.Adam(list(model.parameters()), lr=5e-4)
    it = [20000]
    def step():
        it[0] += 1
        return harness.train_one_iter(model, losses, opt, batcher.get_next_batch(it[0]), sub)
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): step()
    t_host = (time.perf_counter() - t0) / 20
    torch.cuda.synchronize(); t_all = (time.perf_counter() - t0) / 20
    print(f'{rays}+{sparse} rays: host-side enqueue {t_host*1e3:.2f} ms/iter, wall {t_all*1e3:.2f} ms/iter')
