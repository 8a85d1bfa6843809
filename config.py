"""Command-line flags of the WGAN-GP trainer.  Names, types and defaults of the first block are the
reference's (config.py:4-27), so the command line of its README.md:83 works unchanged; the second block
holds this build's own switches, which default to the reference's behaviour."""
import argparse

# (flag, type, default, what it sets)
REFERENCE_FLAGS = (
    ("vocab_size", int, 175, "number of distinct system calls"),
    ("epochs", int, 200, "passes over the training set"),
    ("batch_size", int, 64, "sequences per batch"),
    ("lr", float, 0.0002, "Adam step size of both optimizers"),
    ("b1", float, 0.5, "Adam beta1"),
    ("b2", float, 0.999, "Adam beta2"),
    ("latent_dim", int, 100, "width of the generator's noise input"),
    ("seq_len", int, 256, "sequence length used by the data pipeline"),
    ("n_critic", int, 5, "critic updates per generator update"),
    ("clip_value", float, 0.01, "weight clip of the clipping variant (unused by the gradient-penalty trainer)"),
    ("lambda_gp", float, 10.0, "weight of the gradient penalty"),
    ("hard", bool, False, "straight-through one-hot samples (argparse bool: any non-empty string is True)"),
    ("tau", float, 0.5, "Gumbel-softmax temperature"),
)


def get_args(argv=None):
    parser = argparse.ArgumentParser()
    for name, kind, default, what in REFERENCE_FLAGS:
        parser.add_argument(f"--{name}", type=kind, default=default, help=what)
    # not in the reference
    parser.add_argument("--gen_seq_len", type=int, default=120,
                        help="generator sequence length (the reference hard-codes 120)")
    parser.add_argument("--drop_rate", type=float, default=0.3, help="critic LSTM inter-layer dropout")
    parser.add_argument("--device_rng", action="store_true",
                        help="draw z / Gumbel uniforms / alpha on the GPU (faster; the default draws them on the "
                             "CPU generator in the reference's order, so a seed reproduces the reference's noise)")
    opt = parser.parse_args(argv)
    print(opt)
    return opt
