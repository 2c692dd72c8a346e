"""Mirror of the reference's `lxrt_vis` package: the `lxrt` encoder that additionally hands back
the cross-attention probabilities (`output_attention=True`; reference lxrt_vis/modeling.py:320,347-350,458-462,564-572,
lxrt_vis/entry.py:109-121). Same engine, same state_dict keys; the probabilities are read back from the engine after the
forward pass (`rgqa_engine_get_cross_attention`)."""
