"""Entry point with the reference's usage (`python test.py` -- reference test.py:113-116): load config.load_model, run the
NYU test split through the MI355X inference engine, print the mean joint error and write <work_dir>/test_<mpe>.txt.
Accepts the same `--set key=value` / `--synthetic N` options as train.py."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

if __name__ == "__main__":
    import train
    train.main(test_only=True)
