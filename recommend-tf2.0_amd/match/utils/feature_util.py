"""Feature-column dict schema of the reference (src/match/utils/feature_util.py:1-30)."""


def sparseFeature(feat, feat_num, feat_len=1, embed_dim=4):
    return {'feat': feat, 'feat_num': feat_num, 'feat_len': feat_len, 'embed_dim': embed_dim}


def denseFeature(feat):
    return {'feat': feat}


def varLenSparseFeat(feat, feat_num, maxlen, embed_dim=4):
    return {'feat': feat, 'feat_num': feat_num, 'maxlen': maxlen, 'embed_dim': embed_dim}
