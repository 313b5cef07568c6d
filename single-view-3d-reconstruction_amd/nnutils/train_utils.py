"""Generic trainer with the hooks and loop of the reference's nnutils/train_utils.py
(`Trainer`, train_utils.py:62-256), extended to one process per GPU.

What is kept: the subclass hooks (`define_model`, `init_dataset`, `define_criterion`,
`set_input`, `forward` -> `self.total_loss`, `get_current_scalars`), Adam for `self.model` and
a second Adam for `self.part_basis_generator` when `opts.seg_train and not opts.query_basis`
(:161-172), autocast + GradScaler, unscale -> clip_grad_norm_(0.1) -> step per optimizer
(:201-214), the 0.99/0.01 smoothed loss (:194), `{label}_net_{epoch}.pth` snapshots of the
bare module's state_dict (:84-99) every `save_latest_freq` iterations / `save_epoch_freq`
epochs (:232-253), `num_iter` early stop (:240) and the cut of every epoch after 10 % of the
loader (:243-244, `opts.epoch_fraction`).

What is new (the reference is single-process, `--gpu_id`): the batch shards by image, so
training scales as data parallel replicas.  `init_distributed()` reads RANK / LOCAL_RANK /
WORLD_SIZE (torch.distributed.run), binds the rank to its GPU before any model code runs and
opens the "nccl" (= RCCL over xGMI) process group, or "gloo" on CPU.  `init_training()` wraps
the model(s) in DistributedDataParallel, whose bucketed gradient all-reduce overlaps with the
rasterizer's backward; the rasterizer and Chamfer kernels themselves need no collective.
Clipping runs on the already averaged gradients, so every rank takes the same step.  Rank 0
alone writes snapshots and logs; scalars are averaged over ranks every `print_freq` steps.
BatchNorm statistics stay per rank (no SyncBN), as a single-GPU run of the reference has them
per batch.

`--cuda_graph 1` (single process): a step of the reference's size issues several hundred kernel
launches and is bound by the host issuing them, not by the GPU.  After `GRAPH_EAGER_STEPS`
ordinary steps the trainer captures forward + loss + backward once in a HIP graph
(`torch.cuda.CUDAGraph`) over static copies of the inputs a subclass names in `graph_inputs`, and
from then on replays it; unscale, clipping, the optimizer steps and the scaler update stay eager.
`--cuda_graph 2` captures those too, so that a step is one replay plus the copy of its inputs: it
needs `--fused_adam 1`, whose step takes the scaler's inf check as a device tensor (the unfused
optimizers make GradScaler read that check on the host, which a capture cannot hold).  Everything
in the captured part must be free of host synchronisation and of shapes that change from step to
step.  A captured step also freezes every Python-side decision `forward` took while it was
captured: a subclass whose `forward` branches on the step number (the reference's does --
main.py:299-308: pose / shape warm-up, alternating vertex updates) names what it branches on in
`graph_key(total_steps)`, and the trainer keeps one captured graph per key.

`--fused_adam 1` (default on a GPU): the reference's `torch.optim.Adam(lr, betas=(beta1, 0.999))`
(:165-172) with `fused=True` -- the same update rule in one multi-tensor kernel per optimizer
instead of a dozen, and no host read of the inf check in `GradScaler.step`.
"""
import argparse
import gc
import os
import time

import torch
import torch.distributed as dist


def add_training_flags(parser=None):
    """argparse twin of the reference's absl flags (train_utils.py:27-56)."""
    p = parser or argparse.ArgumentParser()
    p.add_argument('--name', default='exp_name')
    p.add_argument('--gpu_id', type=int, default=0)
    p.add_argument('--num_epochs', type=int, default=1000)
    p.add_argument('--num_pretrain_epochs', type=int, default=0)
    p.add_argument('--learning_rate', type=float, default=0.0001)
    p.add_argument('--beta1', type=float, default=0.9)
    p.add_argument('--use_sgd', action='store_true')
    p.add_argument('--batch_size', type=int, default=4, help='per-process minibatch')
    p.add_argument('--num_iter', type=int, default=0)
    p.add_argument('--checkpoint_dir', default=os.path.join(os.getcwd(), 'snapshots'))
    p.add_argument('--print_freq', type=int, default=10)
    p.add_argument('--save_latest_freq', type=int, default=5000)
    p.add_argument('--save_epoch_freq', type=int, default=25)
    p.add_argument('--print_scalars', type=int, default=1)
    p.add_argument('--seg_train', type=int, default=0)
    p.add_argument('--query_basis', type=int, default=0)
    p.add_argument('--grad_clip', type=float, default=0.1)
    p.add_argument('--epoch_fraction', type=float, default=0.1)
    p.add_argument('--gc_freeze', type=int, default=1,
                   help='freeze the objects alive after set-up out of the garbage collector')
    p.add_argument('--amp', type=int, default=1)
    p.add_argument('--amp_dtype', choices=['fp16', 'bf16'], default='fp16',
                   help='autocast type; the reference uses fp16 with a GradScaler (train_utils.py:182-214)')
    p.add_argument('--seed', type=int, default=0)
    p.add_argument('--cuda_graph', type=int, default=0,
                   help='1: after a few eager steps, capture forward + loss + backward of a step in a HIP graph '
                        'and replay it (single process, constant batch shape; the optimizer steps stay eager); '
                        '2: capture unscale, clipping, the Adam steps and the scaler update as well '
                        '(needs --fused_adam 1)')
    p.add_argument('--fused_adam', type=int, default=1,
                   help="on a GPU, Adam with fused=True: the reference's update rule in one kernel per optimizer")
    return p


class DistEnv(object):
    """Rank bookkeeping of one process."""

    def __init__(self, rank=0, local_rank=0, world_size=1, device=None, backend=None, group_of_one=False):
        self.rank, self.local_rank, self.world_size = rank, local_rank, world_size
        self.device = device if device is not None else torch.device('cpu')
        self.backend = backend
        self.group_of_one = group_of_one        # a process group was opened although there is one rank

    @property
    def is_main(self):
        return self.rank == 0

    @property
    def distributed(self):
        return self.world_size > 1 or self.group_of_one


def init_distributed(backend=None, gpu_id=0, force=False):
    """Open the process group described by the environment (no-op for a single process).
    Must run before the model is built: it selects this rank's GPU.  `force` (or SV3D_FORCE_DIST=1)
    opens the group for a single process as well, so that communicator creation, the DDP wrappers
    and their gradient all-reduce run exactly as they do with several ranks."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', str(gpu_id)))
    backend = backend or os.environ.get('SV3D_DIST_BACKEND') or None
    use_cuda = torch.cuda.is_available() and backend != 'gloo'
    if backend == 'gloo+cuda':
        # rehearsal of the multi-rank step on fewer GPUs than ranks: ranks share the GPUs round
        # robin and gradients travel through gloo (RCCL refuses two ranks on one device)
        use_cuda, backend = torch.cuda.is_available(), 'gloo'
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    if use_cuda:
        if world == 1:
            local_rank = gpu_id
        torch.cuda.set_device(local_rank)
        device = torch.device('cuda', local_rank)
    else:
        device = torch.device('cpu')
    force = bool(force) or os.environ.get('SV3D_FORCE_DIST') == '1'
    grouped = world > 1 or force
    if grouped:
        backend = backend or ('nccl' if use_cuda else 'gloo')
        if not dist.is_initialized():
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            os.environ.setdefault('MASTER_PORT', '29500')
            kwargs = {'device_id': device} if backend == 'nccl' else {}
            dist.init_process_group(backend, rank=rank, world_size=world, **kwargs)
        else:
            backend = dist.get_backend()
    return DistEnv(rank, local_rank, world, device, backend if grouped else None, group_of_one=grouped and world == 1)


def unwrap(module):
    return module.module if isinstance(module, torch.nn.parallel.DistributedDataParallel) else module


class Trainer(object):
    # names of the attributes `set_input` fills that `forward` reads: the inputs of a captured step
    graph_inputs = ()
    GRAPH_EAGER_STEPS = 3

    def __init__(self, opts, env=None):
        self.opts = opts
        self._graph = None              # the captured step in use: {'graph', 'static', 'loss', 'whole', 'key'}
        self._graphs = {}               # (whole, graph_key) -> captured step
        self._graph_static = None       # static copies of the step's inputs, shared by the captured steps
        self._graph_eager_steps = 0
        self.env = env if env is not None else init_distributed(gpu_id=getattr(opts, 'gpu_id', 0))
        self.device = self.env.device
        self.gpu_id = self.env.local_rank if self.device.type == 'cuda' else None
        self.invalid_batch = False      # a subclass may set this in set_input to skip the step
        self.save_dir = os.path.join(opts.checkpoint_dir, opts.name)
        torch.manual_seed(getattr(opts, 'seed', 0) + self.env.rank)
        if self.env.is_main:
            os.makedirs(self.save_dir, exist_ok=True)
            with open(os.path.join(self.save_dir, 'opts.log'), 'w') as f:
                for k, v in sorted(vars(opts).items()):
                    f.write('{}: {}\n'.format(k, v))

    # ---- snapshots (train_utils.py:84-99) ----------------------------------------------
    def save_network(self, network, network_label, epoch_label, gpu_id=None):
        if not self.env.is_main:
            return
        save_path = os.path.join(self.save_dir, '{}_net_{}.pth'.format(network_label, epoch_label))
        state = {k: v.detach().cpu() for k, v in unwrap(network).state_dict().items()}
        torch.save(state, save_path)

    def load_network(self, network, network_label, epoch_label, network_dir=None):
        save_path = os.path.join(network_dir or self.save_dir,
                                 '{}_net_{}.pth'.format(network_label, epoch_label))
        unwrap(network).load_state_dict(torch.load(save_path, map_location=self.device, weights_only=True))

    # ---- hooks ---------------------------------------------------------------------------
    def define_model(self):
        raise NotImplementedError

    def init_dataset(self):
        raise NotImplementedError

    def define_criterion(self):
        raise NotImplementedError

    def set_input(self, batch):
        raise NotImplementedError

    def forward(self, total_steps=0):
        '''Should compute self.total_loss.'''
        raise NotImplementedError

    def get_current_scalars(self):
        return {'total_loss': float(self.total_loss.detach()),
                'smoothed_total_loss': float(self.smoothed_total_loss)}

    def save(self, epoch_prefix):
        self.save_network(self.model, 'pred', epoch_prefix, gpu_id=self.gpu_id)

    def save_basis(self, epoch_prefix):
        self.save_network(self.part_basis_generator, 'part_basis', epoch_prefix, gpu_id=self.gpu_id)

    # ---- set-up --------------------------------------------------------------------------
    def _second_optimizer(self):
        o = self.opts
        return bool(getattr(o, 'seg_train', 0)) and not bool(getattr(o, 'query_basis', 0)) \
            and hasattr(self, 'part_basis_generator')

    def _wrap(self, module):
        module = module.to(self.device)
        if not self.env.distributed or not any(p.requires_grad for p in module.parameters()):
            return module
        ids = [self.env.local_rank] if self.device.type == 'cuda' else None
        return torch.nn.parallel.DistributedDataParallel(module, device_ids=ids, broadcast_buffers=False)

    def make_sampler(self, dataset, shuffle=True):
        """DistributedSampler replacing the reference loader's shuffle=True
        (data/base.py:215-222); keep drop_last=True on the DataLoader."""
        if not self.env.distributed:
            return None
        return torch.utils.data.distributed.DistributedSampler(
            dataset, num_replicas=self.env.world_size, rank=self.env.rank, shuffle=shuffle,
            drop_last=True)

    def init_training(self):
        opts = self.opts
        self.init_dataset()
        self.define_model()
        self.define_criterion()
        self.model = self._wrap(self.model)
        if self._second_optimizer():
            self.part_basis_generator = self._wrap(self.part_basis_generator)

        def make(params):
            if opts.use_sgd:
                return torch.optim.SGD(params, lr=opts.learning_rate, momentum=opts.beta1)
            fused = self.device.type == 'cuda' and bool(getattr(opts, 'fused_adam', 1))
            return torch.optim.Adam(params, lr=opts.learning_rate, betas=(opts.beta1, 0.999),
                                    **({'fused': True} if fused else {}))

        self.optimizer = make(self.model.parameters())
        if self._second_optimizer():
            self.optimizer_2 = make(self.part_basis_generator.parameters())
        if getattr(opts, 'gc_freeze', 1):
            # everything built so far (modules, the imported libraries) lives as long as the run:
            # taken out of the cyclic collector's sight, its full collections stop costing tens of
            # milliseconds every few hundred steps of a step that itself takes a few milliseconds
            gc.collect()
            gc.freeze()

    # ---- loop ------------------------------------------------------------------------------
    def reduce_scalars(self, scalars):
        """Mean over ranks of a dict of python floats (one small all-reduce)."""
        if not self.env.distributed:
            return scalars
        keys = sorted(scalars)
        t = torch.tensor([float(scalars[k]) for k in keys], dtype=torch.float64, device=self.device)
        dist.all_reduce(t)
        t /= self.env.world_size
        return {k: float(v) for k, v in zip(keys, t.tolist())}

    def skip_step(self):
        """The reference skips a step whose `set_input` set `self.invalid_batch`
        (train_utils.py:67,188).  With one process per GPU that decision must be the same on
        every rank -- a rank that skipped alone would leave the others waiting in the gradient
        all-reduce -- so the flag is combined over ranks: all skip if any rank's batch is invalid."""
        if not self.env.distributed:
            return bool(self.invalid_batch)
        flag = torch.tensor([1.0 if self.invalid_batch else 0.0], device=self.device)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        return bool(flag.item() > 0)

    def _zero_grads(self):
        self.optimizer.zero_grad(set_to_none=True)
        if self._second_optimizer():
            self.optimizer_2.zero_grad(set_to_none=True)

    def _forward_backward(self, total_steps, scaler):
        opts = self.opts
        use_amp = self.device.type == 'cuda' and bool(getattr(opts, 'amp', 1))
        amp_dtype = torch.bfloat16 if getattr(opts, 'amp_dtype', 'fp16') == 'bf16' else torch.float16
        with torch.autocast(device_type=self.device.type, dtype=amp_dtype if self.device.type == 'cuda' else None,
                            enabled=use_amp):
            self.forward(total_steps)
        scaler.scale(self.total_loss).backward()

    def _optimizer_steps(self, scaler):
        """unscale -> clip_grad_norm_ -> step for each optimizer, then the scaler update
        (train_utils.py:201-214)."""
        clip = getattr(self.opts, 'grad_clip', 0.1)
        scaler.unscale_(self.optimizer)
        torch.nn.utils.clip_grad_norm_(self.model.parameters(), clip)
        scaler.step(self.optimizer)
        if self._second_optimizer():
            scaler.unscale_(self.optimizer_2)
            torch.nn.utils.clip_grad_norm_(self.part_basis_generator.parameters(), clip)
            scaler.step(self.optimizer_2)
        scaler.update()

    def _graph_wanted(self):
        return int(getattr(self.opts, 'cuda_graph', 0)) > 0 and self.device.type == 'cuda' \
            and not self.env.distributed and len(self.graph_inputs) > 0

    def _graph_whole_step(self):
        return int(getattr(self.opts, 'cuda_graph', 0)) >= 2

    def graph_key(self, total_steps):
        """What `forward`'s Python-side control flow depends on at this step, as a hashable value
        (for the reference's schedule: `(iters < warmup_pose_iter, iters < warmup_shape_iter,
        iters % (2 * warmup_shape_iter) >= warmup_shape_iter)`).  A replayed graph repeats the
        branch it was captured on, so `--cuda_graph` keeps one captured step per key and captures
        again the first time a new key comes up.  The default, None, says that `forward` does not
        look at the step number."""
        return None

    def _norm_state(self):
        """The running statistics of every normalisation layer (what a forward pass in training
        mode changes besides the gradients)."""
        mods = [self.model] + ([self.part_basis_generator] if self._second_optimizer() else [])
        return [b for m in mods for mod in m.modules() if isinstance(mod, torch.nn.modules.batchnorm._NormBase)
                for b in mod.buffers(recurse=False)]

    def _make_optimizers_capturable(self):
        """A captured optimizer step must keep its step counts on the device and take the inf check
        as a tensor: Adam with fused=True does both once its groups are marked capturable."""
        opts = [self.optimizer] + ([self.optimizer_2] if self._second_optimizer() else [])
        for opt in opts:
            if not isinstance(opt, torch.optim.Adam) or not all(g.get('fused') for g in opt.param_groups):
                raise RuntimeError('--cuda_graph 2 captures the optimizer steps and needs Adam with --fused_adam 1 '
                                   '(not --use_sgd); use --cuda_graph 1 to keep the optimizer steps eager')
            for g in opt.param_groups:
                g['capturable'] = True

    def _forward_backward_graphed(self, total_steps, scaler):
        """Forward + loss + backward of this step (and, with `--cuda_graph 2`, the optimizer steps)
        from the captured graph (captured here on first use)."""
        whole = self._graph_whole_step()
        key = (whole, self.graph_key(total_steps))
        self._graph = self._graphs.get(key)       # the mode or the schedule's branch may have changed
        if self._graph_static is None:
            self._graph_static = {name: torch.empty_like(getattr(self, name)) for name in self.graph_inputs}
        static = self._graph_static
        for name, buf in static.items():
            src = getattr(self, name)
            if src.shape != buf.shape or src.dtype != buf.dtype:
                raise RuntimeError('--cuda_graph needs a constant batch: %s is %s, the captured step has %s'
                                   % (name, tuple(src.shape), tuple(buf.shape)))
            buf.copy_(src, non_blocking=True)
            setattr(self, name, buf)
        if self._graph is None:
            if whole:
                self._make_optimizers_capturable()
            # torch's recipe: a few uncaptured passes on a side stream, then the capture; the gradients
            # of these passes are dropped, the weights do not move, and the normalisation layers'
            # running statistics are put back afterwards (the passes would otherwise count this batch
            # twice more).  The last eager step's loss still holds its autograd graph, whose
            # gradient-accumulation nodes belong to the default stream.
            self.total_loss = None
            # (the parameters' accumulation nodes were made on the default stream by the eager steps;
            # torch warns about that once per pass on the side stream, and syncs as needed)
            quiet = getattr(torch.autograd.graph, 'set_warn_on_accumulate_grad_stream_mismatch', None)
            if quiet is not None:
                quiet(False)
            norm = self._norm_state()
            kept = [b.detach().clone() for b in norm]
            # A later capture (another graph_key) must leave the gradients where the first captured
            # step put them: the other graphs write those tensors, and the optimizers read them.
            params = [p for g in self.optimizer.param_groups for p in g['params']]
            if self._second_optimizer():
                params += [p for g in self.optimizer_2.param_groups for p in g['params']]
            held = [p.grad for p in params] if self._graphs else None
            side = torch.cuda.Stream(self.device)
            side.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(side):
                for _ in range(2):
                    self._zero_grads()
                    self._forward_backward(total_steps, scaler)
            torch.cuda.current_stream(self.device).wait_stream(side)
            # nothing made by those passes may be released inside the capture (a tensor with autograd
            # history that was allocated on the side stream and dies during capture ends the capture
            # in a segmentation fault on this PyTorch-ROCm build): their loss goes now
            self.total_loss = None
            with torch.no_grad():
                for b, k in zip(norm, kept):
                    b.copy_(k)
            self._zero_grads()
            if held is not None:
                for p, g in zip(params, held):
                    p.grad = g
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                if held is not None:                # zero in place, as part of the replayed step
                    self.optimizer.zero_grad(set_to_none=False)
                    if self._second_optimizer():
                        self.optimizer_2.zero_grad(set_to_none=False)
                self._forward_backward(total_steps, scaler)
                if whole:
                    self._optimizer_steps(scaler)
            self._graph = {'graph': graph, 'static': static, 'loss': self.total_loss, 'whole': whole, 'key': key}
            self._graphs[key] = self._graph
        self._graph['graph'].replay()
        self.total_loss = self._graph['loss']

    def train_step(self, batch, total_steps, scaler):
        opts = self.opts
        self.set_input(batch)
        if self.skip_step():
            return False
        graphed = self._graph_wanted() and self._graph_eager_steps >= self.GRAPH_EAGER_STEPS
        if graphed:
            self._forward_backward_graphed(total_steps, scaler)
        else:
            self._graph_eager_steps += 1
            self._zero_grads()
            self._forward_backward(total_steps, scaler)
        # the reference's smoothed loss (:194), kept on the device: no host sync per step
        self.smoothed_total_loss = self.smoothed_total_loss * 0.99 + 0.01 * self.total_loss.detach().float()
        if not (graphed and self._graph['whole']):
            self._optimizer_steps(scaler)
        return True

    def train(self):
        opts = self.opts
        self.smoothed_total_loss = 0.0
        dataset_size = len(self.dataloader)
        total_steps = dataset_size * opts.num_pretrain_epochs
        use_amp = self.device.type == 'cuda' and bool(getattr(opts, 'amp', 1))
        if getattr(self, '_scaler', None) is None:      # one scaler per trainer: a captured step holds its scale tensor
            self._scaler = torch.amp.GradScaler(self.device.type, enabled=use_amp)
        scaler = self._scaler
        frac = getattr(opts, 'epoch_fraction', 0.1)
        self.images_seen = 0
        t_log = time.time()
        for epoch in range(opts.num_pretrain_epochs, opts.num_epochs):
            sampler = getattr(self.dataloader, 'sampler', None)
            if hasattr(sampler, 'set_epoch'):
                sampler.set_epoch(epoch)
            epoch_iter = 0
            for i, batch in enumerate(self.dataloader):
                if self.train_step(batch, total_steps, scaler):
                    self.images_seen += opts.batch_size * self.env.world_size
                total_steps += 1
                epoch_iter += 1
                if opts.print_scalars and total_steps % opts.print_freq == 0:
                    # the only host sync: once per print_freq steps (the reference syncs every step)
                    scalars = self.reduce_scalars(self.get_current_scalars())
                    if self.env.is_main:
                        dt = time.time() - t_log
                        rate = opts.print_freq * opts.batch_size * self.env.world_size / max(dt, 1e-9)
                        msg = ', '.join('%s: %.4g' % kv for kv in sorted(scalars.items()))
                        print('(epoch: %d, iters: %d, %.1f img/s over %d rank(s)) %s'
                              % (epoch, epoch_iter, rate, self.env.world_size, msg), flush=True)
                    t_log = time.time()
                if total_steps % opts.save_latest_freq == 0:
                    self.save('latest')
                    if self._second_optimizer():
                        self.save_basis('latest')
                if total_steps == opts.num_iter:
                    return total_steps
                if frac < 1.0 and i / dataset_size > frac:
                    break
            if (epoch + 1) % opts.save_epoch_freq == 0:
                self.save('latest')
                self.save(epoch + 1)
                if self._second_optimizer():
                    self.save_basis('latest')
                    self.save_basis(epoch + 1)
        return total_steps
