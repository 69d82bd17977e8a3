"""Synapse (T-bar) point lists as JSON - the data formats on the output side of the
detection path (reference `flypylib/fplsynapses.py:11-111`) - and the label / mask
volumes training is fed from (`write_labels_mask`, :251-310).  The DVID push / ROI /
annotation-editing helpers of the reference need `libdvid` and are not part of this
package."""
import json
import os

import numpy as np

from . import fplutils


def _raveler_records(doc):
    """{'data': [{'T-bar': {'location': [x, y, z], 'confidence': c}}, ...]}"""
    tbars = [entry['T-bar'] for entry in doc['data']]
    return ([t['location'] for t in tbars], [t['confidence'] for t in tbars], [])


def _dvid_records(elements):
    """[{'Kind': 'PreSyn', 'Pos': [x, y, z], 'Prop': {'conf': '0.9', 'err': ...}}, ...];
    a list holding one such list is unwrapped; other kinds (PostSyn ...) are skipped;
    a missing confidence reads as 1, a missing error estimate as None"""
    if len(elements) == 1 and isinstance(elements[0], list):
        elements = elements[0]
    pre = [e for e in elements if e['Kind'] == 'PreSyn']
    prop = [e['Prop'] for e in pre]
    return ([e['Pos'] for e in pre],
            [float(p['conf']) if 'conf' in p else 1.0 for p in prop],
            [float(p['err']) if 'err' in p else None for p in prop])


def load_from_json(fn, vol_sz=None, buffer=None):
    """T-bars of a json file (or json text) in either of the formats the reference's
    tools exchange (reference fplsynapses.py:11-76): Raveler ({'data': [{'T-bar': ...}]})
    or DVID annotation elements.  With `buffer` (and `vol_sz`) the points closer than
    the buffer to a face of the volume are dropped.
    -> {'locs': (N, 3), 'conf': (N,), 'err': (N,)}"""
    text = open(fn).read() if os.path.isfile(fn) else fn
    doc = json.loads(text)
    if isinstance(doc, dict) and 'data' in doc:
        locs, conf, err = _raveler_records(doc)
    elif doc is None:
        locs, conf, err = [], [], []
    else:
        locs, conf, err = _dvid_records(doc)
    locs, conf, err = np.asarray(locs), np.asarray(conf), np.asarray(err)
    if locs.size > 0 and buffer is not None and buffer != 0:
        assert vol_sz is not None, 'to apply buffer, must also supply volume size'
        lo = np.asarray(fplutils.to3d(buffer))
        hi = np.asarray(fplutils.to3d(vol_sz)) - lo
        inside = np.all((locs >= lo) & (locs < hi), axis=1)
        locs, conf = locs[inside], conf[inside]
    return {'locs': locs, 'conf': conf, 'err': err}


def _dump(obj, json_file):
    if json_file is not None:
        with open(json_file, 'w') as out:
            json.dump(obj, out)
    return obj


def tbars_to_json_format(tbars_np, json_file=None, user_name='$fpl', labels=None):
    """DVID annotation elements of a {'locs', 'conf'} point list (reference :78-96):
    integer positions, the confidence as a '%.03f' string, optionally the body id"""
    positions = np.asarray(tbars_np['locs']).astype('int').tolist()
    elements = [{'Kind': 'PreSyn', 'Pos': pos,
                 'Prop': {'conf': '%.03f' % c, 'user': user_name}}
                for pos, c in zip(positions, np.asarray(tbars_np['conf']).ravel())]
    if labels is not None:
        for element, body in zip(elements, labels):
            element['body ID'] = str(body)
    return _dump(elements, json_file)


def tbars_to_json_format_raveler(tbars_np, json_file=None):
    """Raveler json of a {'locs', 'conf'} point list (reference :98-111)"""
    positions = np.asarray(tbars_np['locs']).astype('int').tolist()
    doc = {'data': [{'T-bar': {'confidence': '%.03f' % c, 'location': pos}}
                    for pos, c in zip(positions, np.asarray(tbars_np['conf']).ravel())]}
    return _dump(doc, json_file)


def write_labels_mask(tbars, roi_mask, radius_use, radius_ign, buffer_size, prefix):
    """training labels and mask around annotated T-bars (reference :251-310): label 1
    within `radius_use` of a T-bar; the mask is cleared in the shell between
    `radius_use` and `radius_ign` (neither positive nor negative) and within
    `buffer_size` of the faces.  Written as the reference's '<prefix>_labels.h5' /
    '<prefix>_mask.h5' (dataset 'main') and as '<prefix>_labels.npy' / '<prefix>_mask.npy',
    and returned."""
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
        from . import keras_io
        np.save('%s_labels.npy' % prefix, labels)
        np.save('%s_mask.npy' % prefix, mask)
        keras_io.write_main('%s_labels.h5' % prefix, labels)
        keras_io.write_main('%s_mask.h5' % prefix, mask)
    return labels, mask
