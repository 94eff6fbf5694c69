"""Mirror of the reference's `match` package surface (src/match/...) on the MI355X HIP kernels."""
