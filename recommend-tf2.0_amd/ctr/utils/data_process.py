"""Mirror of src/ctr/utils/data_process.py for the path: the feature-column dict schema (:13-30) and
`create_criteo_dataset` (:39-91) with the per-row work on the device.

What stays on the host: reading the CSV (pandas) and the one-time vocabulary build of the label encoder.  What moves to
the GPU: LabelEncoder.transform (rec_label_encode_u32), MinMaxScaler fit/transform on astype(int) values
(rec_minmax_*), both bit-identical to scikit-learn (tests/test_pipeline_cpu.py pins the oracle to sklearn,
tests/test_pipeline_gpu.py pins the kernels to the oracle).  Differences from the reference, on purpose:
  * the scaler runs per column (as written the (n, 13) fit_transform result is assigned to one column, which pandas
    rejects: SURVEY §2.1);
  * train_test_split is seeded (`seed`), the reference's is not reproducible.
The Amazon-Electronics and Census loaders (:121-294) are host-only ETL for models outside the path: not mirrored."""
import numpy as np


def sparseFeature(feat, feat_num, embed_dim=4):
    return {'feat': feat, 'feat_num': feat_num, 'embed_dim': embed_dim}


def denseFeature(feat):
    return {'feat': feat}


def create_criteo_dataset(file, embed_dim=8, read_part=True, sample_num=100000, test_size=0.2, seed=2020, device=None):
    """-> feature_columns, (train_X, train_y), (test_X, test_y) exactly as the reference returns them (numpy arrays:
    [dense (n,13) float32, sparse (n,26) int32], labels int32).  `file` is a Criteo CSV with the header
    label,I1..I13,C1..C26, or a pandas DataFrame with those columns."""
    import pandas as pd
    import torch

    from recamd.pipeline import LabelEncoder, MinMaxScaler, hex_tokens
    if isinstance(file, pd.DataFrame):
        data_df = file.iloc[:sample_num] if read_part else file
    elif read_part:
        data_df = pd.read_csv(file, iterator=True).get_chunk(sample_num)            # :54-56
    else:
        data_df = pd.read_csv(file)
    sparse_features = ['C' + str(i) for i in range(1, 27)]
    dense_features = ['I' + str(i) for i in range(1, 14)]
    dev = device or torch.device("cuda", torch.cuda.current_device())
    tokens = np.stack([hex_tokens(data_df[f].tolist()) for f in sparse_features], axis=1)       # fillna('-1') -> MISSING
    dense = np.ascontiguousarray(data_df[dense_features].fillna(0).to_numpy(np.float32))                            # :64
    enc = LabelEncoder(dev).fit(tokens)                                                        # :66-68 (fit)
    ids = enc.transform(torch.from_numpy(tokens.view(np.int32)).to(dev)).cpu().numpy()        # :66-68 (transform)
    scaled = MinMaxScaler().fit_transform(torch.from_numpy(dense).to(dev)).cpu().numpy()      # :76-78, per column
    feature_columns = [[denseFeature(feat) for feat in dense_features]] + \
                      [[sparseFeature(feat, n, embed_dim=embed_dim) for feat, n in zip(sparse_features, enc.classes_)]]
    n = len(data_df)
    perm = np.random.default_rng(seed).permutation(n)                                          # :84 (seeded here)
    n_test = int(np.ceil(n * test_size))
    te, tr = perm[:n_test], perm[n_test:]
    y = data_df['label'].to_numpy().astype('int32')
    return feature_columns, ([scaled[tr], ids[tr]], y[tr]), ([scaled[te], ids[te]], y[te])
