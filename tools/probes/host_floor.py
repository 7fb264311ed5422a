"""Host-side cost of a training iteration (enqueue time with the GPU queue never empty vs wall time), eager and with the
device work of the pass replayed from one HIP graph (harness.GraphedTrainStep), at three batch sizes."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from simplenerf_amd import harness, optim, synth
from simplenerf_amd.data_preprocessors.BatchAssembler01 import BatchAssembler
from simplenerf_amd.loss_functions.LossComputer01 import LossComputer
from simplenerf_amd.models.ModelFactory import get_model
DEV = torch.device('cuda', 0)
prec = os.environ.get('SNERF_PREC', 'f16x3')
for rays, sparse in ((32, 32), (256, 256), (2048, 2048)):
    for mode, binding in (('eager', 'ctypes'), ('eager', 'torch_ext'), ('graph', 'torch_ext'), ('whole', 'torch_ext')):
        cfg = synth.training_configs(prec, rays, sparse); cfg['sub_batch_size'] = rays + sparse
        cfg['model']['hip_host_binding'] = binding
        model = get_model(cfg, None)
        shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
        model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes, 7, 200.0, 8.0).items()})
        model = model.to(DEV).train()
        batcher = BatchAssembler(cfg, synth.training_scene(0, 3, 96, 128, 0.05), DEV)
        losses = LossComputer(cfg); opt = optim.Adam(list(model.parameters()), lr=5e-4)
        it = [20000]
        graphed = harness.GraphedTrainStep(model, losses, batcher.get_next_batch(it[0])) if mode == 'graph' else None
        whole = harness.GraphedIteration(model, losses, opt, batcher) if mode == 'whole' else None
        def step():
            it[0] += 1
            if whole is not None:
                return whole(it[0])
            batch = batcher.get_next_batch(it[0])
            if graphed is None:
                return harness.train_one_iter(model, losses, opt, batch, rays + sparse)
            totals = graphed(batch)
            opt.step()
            return totals
        for _ in range(3): step()
        if whole is not None: whole.wait_seconds = 0.0
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): step()
        t_host = (time.perf_counter() - t0) / 20
        if whole is not None:          # (minus the time it waited for a ring slot: the device was 7 replays behind)
            t_host -= whole.wait_seconds / 20
        torch.cuda.synchronize(); t_all = (time.perf_counter() - t0) / 20
        print(f'{rays}+{sparse} rows {mode:5s} {binding:9s}: host enqueue {t_host*1e3:6.2f} ms/iter, wall {t_all*1e3:6.2f} ms/iter', flush=True)
        del graphed, whole, model, opt
        torch.cuda.empty_cache()
