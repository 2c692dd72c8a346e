from rgqa_amd.butd.preprocess import Dictionary  # noqa: F401
