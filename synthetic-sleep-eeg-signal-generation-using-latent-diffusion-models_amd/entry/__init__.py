"""Entry scripts with the reference's CLI + config/*.yaml surface
(/root/reference/src/train_autoencoderkl.py, train_ldm.py, sample_trials.py), running on libeegldm."""
