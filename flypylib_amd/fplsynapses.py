"""Synapse (T-bar) point lists as JSON - the data formats on the output side of the
detection path (reference `flypylib/fplsynapses.py:11-111`).  The DVID push / ROI /
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
