"""Command-line options of the training entry point (reference: options.py:26-135).

Every reference flag keeps its name, type and default.  Additive flags are grouped under "MI355X" at the end.
"""
import argparse

DATASETS = ['mnist', 'chairs', 'taxibj', 'wave', 'wave_partial', 'sst']
ARCH_TYPES = ['dcgan', 'vgg', 'resnet', 'mlp', 'encoderSST']
DECODER_ARCH_TYPES = ['dcgan', 'vgg', 'mlp', 'decoderSST']
INITIALIZATIONS = ['orthogonal', 'kaiming', 'normal']
MIXING = ['concat', 'mul']


def _build():
    p = argparse.ArgumentParser(prog="PDE-Driven Spatiotemporal Disentanglement (training)",
                                formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    A = p.add_argument
    A('--xp_dir', type=str, metavar='DIR', required=True, help='Directory where models will be saved.')
    A('--chkpt_interval', type=int, metavar='STEPS', default=None,
      help='If not None, save intermediate models every specified number of epochs.')

    g = p.add_argument_group(title='Mixed-precision training',
                             description='Choice of mixed-precision training library.').add_mutually_exclusive_group()
    g.add_argument('--torch_amp', action='store_true',
                   help='Mixed precision as in the reference (torch.cuda.amp): fp16 MFMA operands, fp32 accumulation and master weights, '
                        'dynamic loss scaling.  (--precision bf16 selects the bf16 mode, which needs no scaling.)')
    g.add_argument('--apex_amp', action='store_true', help='Rejected on MI355X (no Apex); kept for CLI compatibility.')

    g = p.add_argument_group(title='Distributed',
                             description='Options for training on GPUs and distributed dataset loading.')
    g.add_argument('--device', type=int, metavar='DEVICE', default=None,
                   help='If not None, indicates the index of the GPU to use.')
    g.add_argument('--num_workers', type=int, metavar='NB', default=4,
                   help='Number of childs processes for data loading.')

    g = p.add_argument_group(title='Model Configuration', description='Model parameters.')
    M = g.add_argument
    M('--nt_cond', type=int, metavar='COND', default=5, help='Number of conditioning observations')
    M('--nt_pred', type=int, metavar='PRED', default=10, help='Number of observations to predict')
    M('--code_size_s', type=int, metavar='SIZE', default=128,
      help='Number of dimensions in S (without skip connections).')
    M('--code_size_t', type=int, metavar='SIZE', default=20, help='Number of dimensions in T.')
    M('--mixing', type=str, metavar='MIXING', default='concat', choices=MIXING,
      help='Whether to concatenate or multiply S and T; in the latter case, their dimensions be equal.')
    M('--architecture', type=str, metavar='ARCH', default='dcgan', choices=ARCH_TYPES,
      help='Encoder and decoder architecture.')
    M('--decoder_architecture', type=str, metavar='ARCH', default=None, choices=DECODER_ARCH_TYPES,
      help='If not None, overwrite the decoder architecture choice.')
    M('--skipco', action='store_true', help='Whether to use skip connections from encoders to decoders.')
    M('--res_hidden_size', type=int, metavar='SIZE', default=512,
      help='Hidden size of MLPs in the residual integrator.')
    M('--n_blocks', type=int, metavar='BLOCKS', default=1, help='Number of resblocks in the residual integrator.')
    M('--enc_hidden_size', type=int, metavar='SIZE', default=64,
      help='Hidden size of MLP encoders, or number of filters in convolutional encoders.')
    M('--dec_hidden_size', type=int, metavar='SIZE', default=64,
      help='Hidden size of MLP decoders, or number of filters in convolutional decoders.')
    M('--enc_n_layers', type=int, metavar='LAYERS', default=3, help='Number of layers in the MLP encoders and decoder.')
    M('--dec_n_layers', type=int, metavar='LAYERS', default=3, help='Number of layers in the MLP encoders and decoder.')
    M('--init_encoder', type=str, metavar='INIT', default='normal', choices=INITIALIZATIONS,
      help='Initialization type of the encoder and the decoder.')
    M('--gain_encoder', type=float, metavar='GAIN', default=0.02,
      help='Initialization gain of the encoder and the decoder.')
    M('--init_resnet', type=str, metavar='INIT', default='orthogonal', choices=INITIALIZATIONS,
      help='Initialization type of the linear layers of the MLP blocks in the integrator.')
    M('--gain_resnet', type=float, metavar='GAIN', default=1.41,
      help='Initialization gain of the linear layers of the MLP blocks in the integrator.')
    M('--no_s', action='store_true', help='If activated, desactivates the static component.')
    M('--offset', type=int, metavar='SIZE', default=5,
      help='When non-zero and equal to the number of conditioning frames, reconstructs conditioning observations, '
           'besides forecasting future observations.')

    g = p.add_argument_group(title='Optimization Configuration', description='Loss and optimization parameters.')
    O = g.add_argument
    O('--lamb_ae', type=float, metavar='LAMBDA', default=10, help='Multiplier of the autoencoding loss.')
    O('--lamb_s', type=float, metavar='LAMBDA', default=45, help='Multiplier of the S invariance loss.')
    O('--lamb_t', type=float, metavar='LAMBDA', default=0.001, help='Multiplier of the T regularization loss.')
    O('--lamb_pred', type=float, metavar='LAMBDA', default=45, help='Multiplier of the prediction loss.')
    O('--batch_size', type=int, metavar='SIZE', default=128, help='Training batch size.')
    O('--lr', type=float, metavar='LR', default=4e-4, help='Learning rate of Adam optimizer.')
    O('--beta1', type=float, metavar='BETA', default=0.9, help='First-order decay parameter of the Adam optimizer.')
    O('--beta2', type=float, metavar='BETA', default=0.99, help='Second-order decay parameter of the Adam optimizer.')
    O('--epochs', type=int, metavar='EPOCH', default=200, help='Number of epochs to train on.')
    O('--scheduler', action='store_true',
      help='If activated, uses a scheluder dividing the learning rate at given epoch milestones.')
    O('--scheduler_decay', type=float, metavar='DECAY', default=0.5,
      help='Multiplier to learning rate applied at each scheduler milestone.')
    O('--scheduler_milestones', type=int, nargs='+', metavar='EPOCHS', default=[300, 400, 500, 600, 700],
      help='Scheduler epoch milestones where the learning rate is multiplied by the decay parameter.')

    g = p.add_argument_group(title='Dataset', description='Chosen dataset and dataset parameters.')
    g.add_argument('--data', type=str, metavar='DATASET', default='mnist', choices=DATASETS, help='Dataset choice.')
    g.add_argument('--data_dir', type=str, metavar='DIR', required=True,
                   help='Data directory; the literal `synthetic` selects seeded synthetic batches of the dataset shape.')
    A('--downsample', type=int, metavar='DOWNSAMPLE', default=2, help='Set the sampling rate for the WaveEq dataset.')
    A('--n_wave_points', type=int, metavar='NUMBER', default=100,
      help='Number of random pixels to select for partial WaveEq (WaveEq-100).')
    A('--zones', type=int, metavar='ZONES', default=list(range(1, 30)), nargs='+', help='SST zones to train on.')
    A('--n_object', type=int, metavar='NUMBER', default=2, help='Number of digits in the Moving MNIST data.')

    g = p.add_argument_group(title='MI355X', description='Additive options of the MI355X-native path.')
    g.add_argument('--precision', type=str, default=None, choices=['fp32', 'bf16', 'fp16'],
                   help='Compute precision of the HIP kernels (default fp32).  fp16 = fp16 MFMA operands + dynamic loss scaling, '
                        'which is what --torch_amp selects (as torch.cuda.amp in the reference); bf16 needs no loss scaling.')
    g.add_argument('--seed', type=int, default=None, help='Seed (the reference draws an unsaved random seed).')
    g.add_argument('--ddp', action='store_true',
                   help='Batch-sharded data parallelism: launch with torchrun, one process per GPU, RCCL all-reduce.')
    g.add_argument('--grad_comm', default='fp32', choices=['fp32', 'bf16'],
                   help='Wire format of the gradient all-reduce with --ddp (bf16 halves the xGMI bytes; fp32 keeps N replicas '
                        'bit-compatible with the single-process step).')
    g.add_argument('--hip_graph', action='store_true',
                   help='(default since round 4; kept so that older command lines still parse) record the whole training step into a '
                        'hipGraph and replay it.')
    g.add_argument('--no_hip_graph', action='store_true',
                   help='Issue every kernel of the step from Python instead of replaying the recorded hipGraph (the WaveEq MLP step is then '
                        'host-bound: ~3.9 ms instead of ~1.4 ms).  The recorded step is what bench.py times; batches of another shape '
                        '(a ragged last batch) run eagerly either way.')
    g.add_argument('--log_interval', type=int, default=None, help='Print losses and frames/s every N steps.')
    g.add_argument('--synthetic_len', type=int, default=2048, help='Sequences per epoch of the synthetic dataset.')
    return p


parser = _build()
