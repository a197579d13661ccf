"""Training entry point: `python -m spatiotemporal_variable_separation_amd.main --xp_dir ... --data_dir ...`
(reference: main.py:49-162, same flags; `--data_dir synthetic` selects seeded synthetic batches)."""
import json
import os

import numpy as np
import torch
import torch.optim.lr_scheduler as lr_scheduler
from torch import optim
from torch.utils.data import DataLoader

from . import functional as VF
from .data.synthetic import SyntheticSequences, data_shape, LAST_ACTIVATION
from .networks.factory import get_encoder, get_decoder, get_resnet
from .networks.model import SeparableNetwork
from .networks.utils import ConstantS
from .options import parser
from .parallel import GradAllReducer, broadcast_module_state
from .train import train


def load_dataset(args, device=None):
    """`--data_dir synthetic` needs nothing.  The WaveEq sets (main.py:91-102 of the reference) are this package's own HBM-resident
    datasets (data/wave_eq.py: batches gathered on the device); Moving MNIST is generated on the device (data/moving_mnist.py).
    Nothing here imports the reference package."""
    if args.data_dir == 'synthetic':
        return SyntheticSequences(args.data, args.nt_cond, args.nt_pred, length=args.synthetic_len,
                                  seed=args.seed or 1234, n_wave_points=args.n_wave_points)
    if args.data in ('wave', 'wave_partial') and device is not None and torch.device(device).type == 'cuda':
        from .data.wave_eq import WaveEq, WaveEqPartial
        if args.data == 'wave':
            return WaveEq(args.data_dir, args.nt_cond, args.nt_cond + args.nt_pred, True, args.downsample, device=device)
        return WaveEqPartial(args.data_dir, args.nt_cond, args.nt_cond + args.nt_pred, True, args.downsample, args.n_wave_points,
                             device=device)
    if args.data == 'mnist' and device is not None and torch.device(device).type == 'cuda':
        from .data.moving_mnist import MovingMNIST                  # sequences generated on the device (data/moving_mnist.py)
        return MovingMNIST.make_dataset(args.data_dir, 64, args.nt_cond, args.nt_cond + args.nt_pred, 4, True, args.n_object, True,
                                        device=device, seed=args.seed)
    raise NotImplementedError(
        'dataset %r: only the synthetic batches (--data_dir synthetic), the WaveEq sets and Moving MNIST are built into this package; '
        'the TaxiBJ / SST / chairs loaders of the reference are host-side file readers (h5py / netCDF4 / image folders) outside the '
        'MI355X hot path -- wrap them in any torch Dataset yielding (cond, target) and call train() directly' % args.data)


def main(argv=None):
    args = parser.parse_args(argv)
    os.environ['OMP_NUM_THREADS'] = str(args.num_workers)
    if not args.ddp:
        from . import configure_single_gpu_queues
        configure_single_gpu_queues()      # before the first HIP call of the process; data-parallel ranks keep the runtime's defaults

    # one process per GPU; under torchrun LOCAL_RANK selects the device
    rank, world = 0, 1
    if args.ddp:
        import torch.distributed as dist
        dist.init_process_group('nccl')
        rank, world = dist.get_rank(), dist.get_world_size()
        args.device = int(os.environ.get('LOCAL_RANK', 0))
    if args.device is None:
        raise RuntimeError('the MI355X-native path has no CPU mode: pass --device N (or --ddp under torchrun)')
    device = torch.device('cuda', args.device)
    torch.cuda.set_device(device)

    seed = np.random.randint(0, 10000) if args.seed is None else args.seed
    if args.ddp:
        from .parallel import broadcast_seed
        seed = broadcast_seed(seed)  # rank 0's draw: without --seed every rank would otherwise draw its own
    torch.manual_seed(seed)
    np.random.seed(seed)            # identical on every rank: one t_random per global step

    if args.data == 'wave_partial':
        assert args.architecture not in ['dcgan', 'vgg']
    shape = data_shape(args.data, args.n_wave_points)
    last_activation = LAST_ACTIVATION[args.data]
    train_set = load_dataset(args, device)
    if args.data == 'wave' and getattr(train_set, 'device_resident', False):
        shape = [1] + list(train_set.frame_shape)      # the reference hard-codes 64x64 (main.py:95); follow the files instead

    if rank == 0:
        os.makedirs(args.xp_dir, exist_ok=True)
        with open(os.path.join(args.xp_dir, 'params.json'), 'w') as f:
            json.dump(vars(args), f, indent=4, sort_keys=True)

    def worker_init_fn(worker_id):
        np.random.seed((torch.randint(100000, []).item() + worker_id))
    sampler = None
    if world > 1:
        from torch.utils.data.distributed import DistributedSampler
        sampler = DistributedSampler(train_set, num_replicas=world, rank=rank, shuffle=True, seed=seed)
    if getattr(train_set, 'generated_on_device', False):
        # Moving MNIST: every item is a fresh video, indices mean nothing -- no sampler.  The draws come from the global NumPy
        # stream like the reference's; with several ranks that stream is shared (it also draws t_random and must agree across
        # replicas), so each rank gets a private stream for its videos and 1/world of the epoch
        from .data.moving_mnist import DeviceMovingLoader
        if world > 1:
            train_set.rng = np.random.RandomState(seed + 7919 * (rank + 1))
        epoch_len = int(os.environ['VARSEP_MMNIST_EPOCH_LEN']) if os.environ.get('VARSEP_MMNIST_EPOCH_LEN') else None   # default 200000 (reference)
        train_loader = DeviceMovingLoader(train_set, args.batch_size, world=world, epoch_len=epoch_len)
    elif getattr(train_set, 'device_resident', False):
        from .data.wave_eq import DeviceBatchLoader                # same sampler stream as the DataLoader below, one gather launch per batch
        train_loader = DeviceBatchLoader(train_set, args.batch_size, shuffle=sampler is None, sampler=sampler)
    else:
        train_loader = DataLoader(train_set, batch_size=args.batch_size, pin_memory=True, shuffle=sampler is None,
                                  sampler=sampler, num_workers=args.num_workers, worker_init_fn=worker_init_fn)

    if not args.no_s:
        Es = get_encoder(args.architecture, shape, args.code_size_s, args.enc_hidden_size, args.enc_n_layers,
                         args.nt_cond, args.init_encoder, args.gain_encoder).to(device)
    else:
        assert not args.skipco
        args.code_size_s = args.code_size_t
        args.mixing = 'mul'
        Es = ConstantS(return_value=1, code_size=args.code_size_s).to(device)
    Et = get_encoder(args.architecture, shape, args.code_size_t, args.enc_hidden_size, args.enc_n_layers,
                     args.nt_cond, args.init_encoder, args.gain_encoder).to(device)
    decoder = get_decoder(args.architecture if args.decoder_architecture is None else args.decoder_architecture,
                          shape, args.code_size_t, args.code_size_s, last_activation, args.dec_hidden_size,
                          args.dec_n_layers, args.mixing, args.skipco, args.init_encoder, args.gain_encoder).to(device)
    t_resnet = get_resnet(args.code_size_t, args.n_blocks, args.res_hidden_size, args.init_resnet, args.gain_resnet,
                          args.architecture == 'encoderSST').to(device)
    sep_net = SeparableNetwork(Es, Et, t_resnet, decoder, args.nt_cond, args.skipco)

    grad_sync = None
    if world > 1:
        broadcast_module_state(sep_net)
        lowp = getattr(args, 'grad_comm', 'fp32') == 'bf16'
        from .train import _mlp_family, chain_weight_parameters, rollout_weight_stacks, shard_optimizer_default
        direct = chain_weight_parameters(sep_net) if (lowp and not getattr(args, 'no_hip_graph', False)) else None
        # MLP family, bf16 wire, recorded step: reduce-scatter + Adam on this rank's slice + all-gather of the operand copies
        # (parallel.GradAllReducer, "Sharded optimizer"); GraphedStep falls back to the all-reduce when the precision mode does not allow it
        grad_sync = GradAllReducer(sep_net.parameters(), comm_dtype=torch.bfloat16 if lowp else torch.float32,
                                   early=None if _mlp_family(sep_net) else list(decoder.parameters()),
                                   lowp_direct=direct, shard_direct=bool(direct) and shard_optimizer_default(),
                                   stacked=rollout_weight_stacks(sep_net))

    # same constructor call as the reference (main.py:133); the update runs as one multi-tensor HIP launch (optim.py)
    from .optim import Adam
    optimizer = Adam(sep_net.parameters(), lr=args.lr, betas=(args.beta1, args.beta2))
    scheduler = lr_scheduler.MultiStepLR(optimizer, args.scheduler_milestones, gamma=args.scheduler_decay) \
        if args.scheduler else None

    # --torch_amp keeps the reference's meaning (fp16 autocast + GradScaler, train.py:96-97); --precision names a mode explicitly
    precision = args.precision or ('fp16' if args.torch_amp else 'fp32')
    VF.set_precision(precision)
    # the recorded step (train.GraphedStep) is the default: main always builds the HIP Adam (recordable, maintains the 16-bit operand
    # copies inside the recording) and its loaders yield one batch shape per epoch but for a ragged last batch, which runs eagerly
    hip_graph = not getattr(args, 'no_hip_graph', False)
    if rank == 0:
        print('compute precision: %s%s' % (precision, ' + dynamic loss scaling' if precision == 'fp16' else
                                            (' (the reference\'s default arithmetic; --precision bf16 selects the 16-bit MFMA kernels bench.py times)'
                                             if precision == 'fp32' else '')))
        print('step launch: %s' % ('recorded hipGraph, replayed per batch (--no_hip_graph: eager launches)' if hip_graph else 'eager launches from Python'))
    from .train import make_loss_scaler
    scaler = make_loss_scaler(device) if precision == 'fp16' else None
    from .train import enable_update_in_backward
    enable_update_in_backward(optimizer, sep_net, grad_sync, scaler=scaler)
    train(args.xp_dir, train_loader, device, sep_net, optimizer, scheduler, args.apex_amp, False, args.epochs, args.lamb_ae,
          args.lamb_s, args.lamb_t, args.lamb_pred, args.offset, args.nt_cond, args.nt_pred, args.no_s, args.skipco,
          args.chkpt_interval, args.architecture == 'encoderSST', grad_sync=grad_sync, log_interval=args.log_interval,
          hip_graph=hip_graph, scaler=scaler)


if __name__ == "__main__":
    main()
