"""Command-line flags of the WGAN-GP trainer: same names, types and defaults as the reference's
config.py:4-27 (so README.md:83's command line works unchanged), plus build-specific extras
that default to the reference's behaviour."""
import argparse


def get_args(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument("--vocab_size", type=int, default=175)
    parser.add_argument("--epochs", type=int, default=200, help="number of epochs of training")
    parser.add_argument("--batch_size", type=int, default=64, help="size of the batches")
    parser.add_argument("--lr", type=float, default=0.0002, help="adam: learning rate")
    parser.add_argument("--b1", type=float, default=0.5, help="adam: decay of first order momentum of gradient")
    parser.add_argument("--b2", type=float, default=0.999, help="adam: decay of first order momentum of gradient")
    parser.add_argument("--latent_dim", type=int, default=100, help="dimensionality of the latent space")
    parser.add_argument("--seq_len", type=int, default=256, help="length of each sequence dimension")
    parser.add_argument("--n_critic", type=int, default=5, help="number of training steps for discriminator per iter")
    parser.add_argument("--clip_value", type=float, default=0.01, help="lower and upper clip value for disc. weights")
    parser.add_argument("--lambda_gp", type=float, default=10.0)
    parser.add_argument("--hard", type=bool, default=False)   # argparse bool: any non-empty string is True
    parser.add_argument("--tau", type=float, default=0.5)
    # extras (not in the reference)
    parser.add_argument("--gen_seq_len", type=int, default=120,
                        help="generator sequence length (the reference hard-codes 120)")
    parser.add_argument("--drop_rate", type=float, default=0.3, help="critic LSTM inter-layer dropout")
    parser.add_argument("--device_rng", action="store_true",
                        help="draw z / Gumbel uniforms / alpha on the GPU (faster; the default draws them on the "
                             "CPU generator in the reference's order, so a seed reproduces the reference's noise)")

    opt = parser.parse_args(argv)
    print(opt)
    return opt
