"""Synapse (T-bar) point lists as JSON - the data formats on the output side of the
detection path (reference `flypylib/fplsynapses.py:11-111`) - and the label / mask
volumes training is fed from (`write_labels_mask`, :251-310).  The DVID push / ROI /
annotation-editing helpers of the reference need `libdvid` and are not part of this
package."""
import json
import os

import numpy as np

from . import fplutils


def load_from_json(fn, vol_sz=None, buffer=None):
    """read T-bars from a json file (or json text): Raveler format
    {'data': [{'T-bar': {'location', 'confidence'}}]} or DVID annotation elements
    [{'Kind': 'PreSyn', 'Pos', 'Prop': {'conf', 'err'}}].  With `buffer` (and
    `vol_sz`), points within the buffer of the volume faces are dropped.
    -> {'locs': (N,3), 'conf': (N,), 'err': (N,)} (reference :11-76)"""
    if os.path.isfile(fn):
        with open(fn) as json_file:
            data = json.load(json_file)
    else:
        data = json.loads(fn)
    locs, conf, err = [], [], []
    if isinstance(data, dict) and 'data' in data.keys():        # Raveler format
        for syn in data['data']:
            locs.append(syn['T-bar']['location'])
            conf.append(syn['T-bar']['confidence'])
    elif data is not None:                                       # DVID format
        if len(data) == 1 and isinstance(data[0], list):
            data = data[0]
        for syn in data:
            if syn['Kind'] != 'PreSyn':
                continue
            conf.append(float(syn['Prop']['conf']) if 'conf' in syn['Prop'] else 1.0)
            err.append(float(syn['Prop']['err']) if 'err' in syn['Prop'] else None)
            locs.append(syn['Pos'])
    locs, conf, err = np.asarray(locs), np.asarray(conf), np.asarray(err)
    if locs.size > 0 and buffer is not None and buffer != 0:
        assert vol_sz is not None, 'to apply buffer, must also supply volume size'
        buffer = fplutils.to3d(buffer)
        vol_sz = fplutils.to3d(vol_sz)
        drop = np.zeros(locs.shape[0], bool)
        for a in range(3):
            drop |= (locs[:, a] < buffer[a]) | (locs[:, a] >= vol_sz[a] - buffer[a])
        locs, conf = locs[~drop], conf[~drop]
    return {'locs': locs, 'conf': conf, 'err': err}


def tbars_to_json_format(tbars_np, json_file=None, user_name='$fpl', labels=None):
    """DVID annotation elements for a {'locs','conf'} point list (reference :78-96)"""
    tbars_json = []
    locs, conf = tbars_np['locs'], tbars_np['conf']
    for ii in np.arange(conf.size):
        tt = {'Kind': 'PreSyn', 'Pos': locs[ii, :].astype('int').tolist(),
              'Prop': {'conf': '%.03f' % conf[ii], 'user': user_name}}
        if labels is not None:
            tt['body ID'] = str(labels[ii])
        tbars_json.append(tt)
    if json_file is not None:
        with open(json_file, 'w') as f_out:
            json.dump(tbars_json, f_out)
    return tbars_json


def tbars_to_json_format_raveler(tbars_np, json_file=None):
    """Raveler-format json for a {'locs','conf'} point list (reference :98-111)"""
    locs, conf = tbars_np['locs'], tbars_np['conf']
    tbars_json = {'data': [{'T-bar': {'confidence': '%.03f' % conf[ii],
                                      'location': locs[ii, :].astype('int').tolist()}}
                           for ii in np.arange(conf.size)]}
    if json_file is not None:
        with open(json_file, 'w') as f_out:
            json.dump(tbars_json, f_out)
    return tbars_json


def write_labels_mask(tbars, roi_mask, radius_use, radius_ign, buffer_size, prefix):
    """training labels and mask around annotated T-bars (reference :251-310): label 1
    within `radius_use` of a T-bar; the mask is cleared in the shell between
    `radius_use` and `radius_ign` (neither positive nor negative) and within
    `buffer_size` of the faces.  Written as '<prefix>_labels.npy' / '<prefix>_mask.npy'
    (the reference writes .h5; h5py is not available here) and returned."""
    radius_use_flt = fplutils.set_filter(radius_use)
    if radius_ign is not None:
        radius_ign_flt = 1 - fplutils.set_filter(radius_ign)
    else:
        radius_ign = 0
    mask = np.copy(roi_mask)
    labels = np.zeros(mask.shape, dtype='uint8')
    for jj in range(tbars['locs'].shape[0]):
        xx, yy, zz = (int(v) for v in tbars['locs'][jj, :3])
        if radius_ign > 0:
            box = (slice(zz - radius_ign, zz + radius_ign + 1),
                   slice(yy - radius_ign, yy + radius_ign + 1),
                   slice(xx - radius_ign, xx + radius_ign + 1))
            mask[box] = np.logical_and(mask[box], radius_ign_flt)
        box = (slice(zz - radius_use, zz + radius_use + 1),
               slice(yy - radius_use, yy + radius_use + 1),
               slice(xx - radius_use, xx + radius_use + 1))
        mask[box] = np.logical_or(mask[box], radius_use_flt)
        labels[box] = np.logical_or(labels[box], radius_use_flt)
    for ax in range(3):
        sl = [slice(None)] * 3
        sl[ax] = slice(0, buffer_size); mask[tuple(sl)] = 0
        sl[ax] = slice(-buffer_size, None); mask[tuple(sl)] = 0
    if prefix is not None:
        np.save('%s_labels.npy' % prefix, labels)
        np.save('%s_mask.npy' % prefix, mask)
    return labels, mask
