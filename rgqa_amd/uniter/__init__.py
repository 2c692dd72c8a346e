"""Mirror of the reference's `uniter` package: the UNITER single-stream backbone for GQA -
`uniter.uniter.GQAUNITER(num_answers)`, `uniter.entry.UniterEncoder`, `uniter.modeling.UniterFeatureExtraction` - on the same HIP
engine (`arch = 2`): 12 BertLayers over one [20 text tokens ; 36 regions] sequence per sample, identical state_dict keys."""
