"""rgqa_amd: the LXMERT-GQA train-step hot path on MI355X."""
