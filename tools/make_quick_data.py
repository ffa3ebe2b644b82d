#!/usr/bin/env python3
"""Write the preprocessed-epoch files reviews4rec_amd.data_fast reads, from a dataset's pickles
(counterpart of the reference's data_scripts/make_quick_data.py: same eight gzip datasets a..h, i8 / f8,
same directory layout -- quick_data_deepconn/ or quick_data_narre/ + <dataset>/<k>_core/ -- and the same
container: train.hdf5 / test.hdf5 / val.hdf5, written by reviews4rec_amd.hdf5_lite (no h5py here) and
readable by the reference's own h5py loader; pass `npz` as the last argument for .npz files instead).

    python tools/make_quick_data.py <dataset> <k_core> <percent> <model_type> [data_root=data/] [hdf5|npz]

Needed only for hyper_params['loader'] = 'fast'; the default loader (reviews4rec_amd/data.py) builds
batches on the device straight from the pickles.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def write_split(reader, path):
    from reviews4rec_amd.data_fast import save_split
    parts, ys = [[] for _ in range(7)], []
    for data, y in reader.iter_review(simple=True):          # data.py:282-291, make_quick_data.py:35
        for slot, d in zip(parts, data):
            slot.append(np.asarray(d))
        ys.append(np.asarray(y))
    if not ys:
        raise SystemExit('empty split: ' + path)
    save_split(path, [np.concatenate(p) for p in parts], np.concatenate(ys))


def main(argv):
    from reviews4rec_amd.data import load_data
    dataset, k_core, percent, model_type = argv[1], int(argv[2]), int(argv[3]), argv[4]
    data_root = argv[5] if len(argv) > 5 else 'data/'
    hp = {'dataset': dataset, 'k_core': k_core, 'percent_reviews_to_keep': percent, 'input_length': 1000,
          'model_type': model_type, 'narre_num_reviews': 10, 'narre_num_words': 100, 'batch_size': 4096}
    rel = dataset + '/' + str(k_core) + '_core/' + (str(percent) + '_percent/' if percent != 100 else '')
    hp['data_dir'] = data_root + rel
    train, test, val, hp = load_data(hp, load_negs=False, device='cpu')
    out = ('quick_data_narre/' if model_type == 'NARRE' else 'quick_data_deepconn/') + rel
    ext = '.' + (argv[6] if len(argv) > 6 else 'hdf5')
    if ext not in ('.hdf5', '.npz'):
        raise SystemExit('format %r: hdf5 or npz' % ext[1:])
    for name, reader in (('train', train), ('test', test), ('val', val)):
        write_split(reader, out + name + ext)
        print('wrote', out + name + ext, len(reader.data), 'ratings')


if __name__ == '__main__':
    main(sys.argv)
