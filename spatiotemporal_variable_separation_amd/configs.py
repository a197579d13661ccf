"""The five BASELINE.json configurations as option dictionaries (README.md recipes of the reference), plus the chairs recipe
(README.md:78) as an extra bench / parity workload."""

BASELINE_CONFIGS = {
    # configs[0]: Moving MNIST 64x64, nt_cond=5 nt_pred=10, batch 16, DCGAN (options.py defaults)
    'mnist_b16': dict(data='mnist', architecture='dcgan', shape=[1, 64, 64], nt_cond=5, nt_pred=10, offset=5,
                      batch=16, code_size_s=128, code_size_t=20, enc_hidden_size=64, dec_hidden_size=64,
                      res_hidden_size=512, n_blocks=1, mixing='concat', last_activation='sigmoid', skipco=False,
                      gain_resnet=1.41, lambdas=dict(ae=10.0, s=45.0, t=0.001, pred=45.0)),
    # configs[1]: WaveEq MLP (README.md:90)
    'waveeq': dict(data='wave', architecture='mlp', shape=[1, 64, 64], nt_cond=5, nt_pred=20, offset=5, batch=128,
                   code_size_s=32, code_size_t=32, enc_hidden_size=1200, dec_hidden_size=1200, enc_n_layers=3,
                   dec_n_layers=4, res_hidden_size=512, n_blocks=3, mixing='mul', last_activation='sigmoid',
                   skipco=False, gain_resnet=0.71, lambdas=dict(ae=1.0, s=45.0, t=0.001, pred=45.0)),
    # configs[2]: Moving MNIST DCGAN, batch 128
    'mnist_b128': dict(data='mnist', architecture='dcgan', shape=[1, 64, 64], nt_cond=5, nt_pred=10, offset=5,
                       batch=128, code_size_s=128, code_size_t=20, enc_hidden_size=64, dec_hidden_size=64,
                       res_hidden_size=512, n_blocks=1, mixing='concat', last_activation='sigmoid', skipco=False,
                       gain_resnet=1.41, lambdas=dict(ae=10.0, s=45.0, t=0.001, pred=45.0)),
    # configs[3]: TaxiBJ VGG32 (README.md:82), batch 100 per GPU
    'taxibj': dict(data='taxibj', architecture='vgg', shape=[2, 32, 32], nt_cond=4, nt_pred=4, offset=4, batch=100,
                   code_size_s=128, code_size_t=20, enc_hidden_size=64, dec_hidden_size=64, res_hidden_size=512,
                   n_blocks=1, mixing='concat', last_activation=None, skipco=False, gain_resnet=0.71,
                   lambdas=dict(ae=45.0, s=0.0001, t=0.001, pred=45.0)),
    # configs[4]: SST (README.md:86) with nt_pred=40, batch 8 per GPU
    'sst': dict(data='sst', architecture='encoderSST', decoder_architecture='decoderSST', shape=[1, 64, 64],
                nt_cond=4, nt_pred=40, offset=0, batch=8, code_size_s=196, code_size_t=64, enc_hidden_size=64,
                dec_hidden_size=64, res_hidden_size=512, n_blocks=2, mixing='concat', last_activation=None,
                skipco=True, average_tloss=True, gain_resnet=0.71, lambdas=dict(ae=1.0, s=100.0, t=5e-6, pred=45.0)),
    # not in BASELINE.json: 3D Warehouse chairs (README.md:78): ResNet18 encoders + DCGAN decoder, options.py defaults otherwise
    'chairs': dict(data='chairs', architecture='resnet', decoder_architecture='dcgan', shape=[3, 64, 64], nt_cond=5, nt_pred=10,
                   offset=5, batch=128, code_size_s=128, code_size_t=10, enc_hidden_size=64, dec_hidden_size=64,
                   res_hidden_size=512, n_blocks=1, mixing='concat', last_activation='sigmoid', skipco=False, gain_resnet=0.71,
                   lambdas=dict(ae=1.0, s=1.0, t=0.001, pred=45.0)),
}
