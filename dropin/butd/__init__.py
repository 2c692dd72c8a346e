"""Drop-in `butd` package: butd.butd.GQABUTD and butd.preprocess.Dictionary resolve to rgqa_amd.butd.*."""
