"""Mirror of the reference's `ctr` package surface (src/ctr/...) on the MI355X HIP kernels.
Same import paths, class names, constructor keywords and `call(inputs)` structure as the reference
(e.g. `from ctr.layers.modules import FM`, `from ctr.deep_fm.model import DeepFM`)."""
