"""Noise floor of the recorded data-parallel MLP step (world size 1, RCCL): two identical replicated runs against each other, and the
sharded-optimizer run against one of them (max |difference| of parameters and exp_avg per tensor)."""
import os, sys, tempfile
import torch
import torch.multiprocessing as mp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))


def main():
    import test_ddp_gpu as T
    d = tempfile.mkdtemp()
    runs = {}
    for name, shard in (('repl_a', False), ('repl_b', False), ('shard', True)):
        out = os.path.join(d, name)
        os.makedirs(out)
        mp.spawn(T._shard_worker, args=(1, T._free_port(), 'nccl', shard, out), nprocs=1, join=True)
        runs[name] = torch.load(os.path.join(out, 'rank0.pt'))
    for key in ('state', 'moments'):
        print(key)
        for k in runs['repl_a'][key]:
            a, b, s = runs['repl_a'][key][k].float(), runs['repl_b'][key][k].float(), runs['shard'][key][k].float()
            print(f'  {k:36s} repl-repl {(a - b).abs().max().item():.3e}   shard-repl {(s - a).abs().max().item():.3e}   max |v| {a.abs().max().item():.3e}')
    print('losses', runs['repl_a']['losses'], runs['repl_b']['losses'], runs['shard']['losses'])


if __name__ == '__main__':
    main()
