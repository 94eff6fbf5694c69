"""Feature-column dict schema of the reference (src/ctr/utils/data_process.py:13-30).  Only the
schema is part of the hot path's API surface; the pandas/sklearn dataset loaders are out of scope."""


def sparseFeature(feat, feat_num, embed_dim=4):
    return {'feat': feat, 'feat_num': feat_num, 'embed_dim': embed_dim}


def denseFeature(feat):
    return {'feat': feat}
