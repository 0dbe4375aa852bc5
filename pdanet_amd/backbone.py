"""IASSD_Backbone restated: builds the six SA layers of PDA-SSD from the (unchanged) yaml and
runs them (/root/reference/pcdet/models/backbones_3d/IASSD_backbone.py:9-240).

Same constructor signature, forward(batch_dict) contract, output keys and state-dict keys
(`SA_modules.<k>....`).  Layer classes are chosen by index exactly as the reference does
(:62-94): k in {1,2,3} -> the PDA layer, k = 0 or k > 4 -> the vanilla SA layer, and the
`Vote_Layer` entry -> Vote_layer.

Differences (results unchanged): the per-scene point count is taken from the tensor shape
(the reference counts `batch_idx == b` per scene and asserts min == max, a device->host sync
per forward, :131-137), and the D-FPS of layer k+1 can be launched on a side stream when it
depends only on coordinates (see `prefetch_fps`).
"""
import contextlib
import os

import torch
import torch.nn as nn

from . import pointnet2_modules, pointnet2_utils


class IASSD_Backbone(nn.Module):
    def __init__(self, model_cfg, num_class, input_channels, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        self.num_class = num_class
        self.SA_modules = nn.ModuleList()
        channel_in = input_channels - 3
        channel_out_list = [channel_in]
        self.num_points_each_layer = []

        sa_config = self.model_cfg.SA_CONFIG
        self.layer_types = sa_config.LAYER_TYPE
        self.ctr_idx_list = sa_config.CTR_INDEX
        self.layer_inputs = sa_config.LAYER_INPUT
        self.aggregation_mlps = sa_config.get('AGGREGATION_MLPS', None)
        self.confidence_mlps = sa_config.get('CONFIDENCE_MLPS', None)
        self.max_translate_range = sa_config.get('MAX_TRANSLATE_RANGE', None)

        for k in range(len(sa_config.NSAMPLE_LIST)):
            if isinstance(self.layer_inputs[k], list):
                channel_in = channel_out_list[self.layer_inputs[k][-1]]
            else:
                channel_in = channel_out_list[self.layer_inputs[k]]

            if self.layer_types[k] == 'SA_Layer':
                mlps = [[channel_in] + list(spec) for spec in sa_config.MLPS[k]]
                channel_out = sum(spec[-1] for spec in mlps)
                aggregation_mlp = None
                if self.aggregation_mlps and self.aggregation_mlps[k]:
                    aggregation_mlp = list(self.aggregation_mlps[k])
                    channel_out = aggregation_mlp[-1]
                confidence_mlp = None
                if self.confidence_mlps and self.confidence_mlps[k]:
                    confidence_mlp = list(self.confidence_mlps[k])
                cls = (pointnet2_modules.PointnetSAModuleMSG_WithSampling if (k < 1 or k > 4)
                       else pointnet2_modules.PointnetSAModuleMSG_WithSampling_Ellipsoid)
                self.SA_modules.append(cls(
                    npoint_list=sa_config.NPOINT_LIST[k],
                    sample_range_list=sa_config.SAMPLE_RANGE_LIST[k],
                    sample_type_list=sa_config.SAMPLE_METHOD_LIST[k],
                    radii=sa_config.RADIUS_LIST[k],
                    nsamples=sa_config.NSAMPLE_LIST[k],
                    mlps=mlps,
                    use_xyz=True,
                    dilated_group=sa_config.DILATED_GROUP[k],
                    aggregation_mlp=aggregation_mlp,
                    confidence_mlp=confidence_mlp,
                    num_class=self.num_class))
            elif self.layer_types[k] == 'Vote_Layer':
                self.SA_modules.append(pointnet2_modules.Vote_layer(
                    mlp_list=sa_config.MLPS[k],
                    pre_channel=channel_out_list[self.layer_inputs[k]],
                    max_translate_range=self.max_translate_range))
            channel_out_list.append(channel_out)
        self.num_point_features = channel_out
        # MI355X: D-FPS uses one CU per scene and depends on coordinates only, so the leading
        # D-FPS layers are sampled on a side stream while the main stream runs layer 0.
        self.prefetch_sampling = True
        self._side_stream = None
        self._prefetched = []          # FIFO of (key, presampled) started by prefetch(), oldest first, at most two
        # Inference (eval, no_grad): the layers behind the last host read of a token count (first_static_tail_layer: ONCE
        # layers 3, 4, 5 -- static shapes, ~100 launches) replayed as ONE hipGraph.  The tensors they return are the graph's
        # own buffers: valid until the next forward.  Off by default; bench.py's inference workload switches it on.
        self.graph_tail_infer = False
        self._tail_graph = None

    def _presample(self, xyz, points=None, batch_size=None, limit=None):
        """Sampling of the leading layers that need only coordinates (identity / D-FPS, chained
        inputs, no given centres), issued on a side stream.  Returns {layer: (event, idx, new_xyz)}.

        `points` given (batch_dict['inputs_resident'] = True): the caller states that the batch already
        sits in HBM (no copy into it is still in flight on the current stream).  The side stream then
        slices its own xyz out of `points` and does NOT wait for the current stream, so in a training or
        serving loop, where the host runs ahead of the device, the D-FPS of step i+1 (a 4095-round latency
        chain on 2 CUs) starts as soon as it is enqueued, under the tail of step i, instead of behind it."""
        plan = []
        for i in range(len(self.SA_modules)):
            m = self.SA_modules[i]
            if self.layer_types[i] != 'SA_Layer' or self.ctr_idx_list[i] != -1 or self.layer_inputs[i] != i:
                break
            if not pointnet2_modules.coordinate_only_sampling(m.sample_type_list, m.sample_range_list, m.npoint_list):
                break
            if limit is not None and i >= limit:      # layers from `limit` on are replayed as a graph and sample inside it
                break
            plan.append(i)
        if not plan or not xyz.is_cuda:
            return {}
        if xyz.shape[1] > 24576 and os.environ.get("PDA_COOP_FPS_SIDE_STREAM", "1") == "0":
            # The multi-workgroup FPS form (csrc/fps.hip, n > 24576) exchanges winners between the K workgroups of a scene.
            # On the side stream its workgroups share the chip with the main stream's kernels and may become resident late;
            # that cannot deadlock (the main stream's workgroups finish on their own and free their CUs; every poll is
            # bounded; a time-out is recovered on the device and counted: pda_fps_coop_timeouts).  Config 5 (65536 pts,
            # B = 8): 32 of 256 CUs busy for 33 ms when it runs in program order, hidden under the previous iteration's
            # backward here: 105 -> 73 ms per iteration, no time-out in any run.  PDA_COOP_FPS_SIDE_STREAM=0: program order.
            return {}
        if self._side_stream is None or self._side_stream.device != xyz.device:
            self._side_stream = torch.cuda.Stream(device=xyz.device)
        main = torch.cuda.current_stream(xyz.device)
        side = self._side_stream
        if points is None:
            side.wait_stream(main)
        out = {}
        cur = xyz
        with torch.cuda.stream(side), torch.no_grad():
            if points is not None:
                points.record_stream(side)
                cur = points[:, 1:4].reshape(batch_size, -1, 3).contiguous()
            for i in plan:
                m = self.SA_modules[i]
                idx = pointnet2_modules.sample_points(cur, None, None, m.sample_type_list, m.sample_range_list,
                                                      m.npoint_list)
                new_xyz = pointnet2_modules.pointnet2_utils.gather_operation(
                    cur.transpose(1, 2).contiguous(), idx).transpose(1, 2).contiguous()
                # a PDA layer's neighbour lists (and distinct-token plans) depend on coordinates only as well
                pre = m.prequery(cur, new_xyz) if hasattr(m, "prequery") and len(m.groupers) > 0 else None
                ev = torch.cuda.Event()
                ev.record(side)
                keep = [idx, new_xyz, cur]
                if pre is not None:
                    keep += list(pre['idxs'])
                    for part in (pre['parts'] or []):
                        # every device tensor of the plan (cnt, off, rowmap AND roww): all are allocated on this side stream,
                        # read by main-stream kernels and some saved for backward
                        keep += [t for t in part if torch.is_tensor(t)]
                for t in keep:
                    t.record_stream(main)
                out[i] = (ev, idx, new_xyz, pre)
                cur = new_xyz
        return out

    def prefetch(self, points, batch_size):
        """Start the coordinate-only front of the NEXT forward now (on the side stream): sampling of the leading layers,
        their ball queries and unique-token plans.  `points` must be the tensor the next forward is called with, already
        resident in HBM and not modified in between.  In a training loop this is called right after the forward of step i
        with the batch of step i+1 (a data loader's prefetch): the 3.7 ms D-FPS latency chain and the plan's host read then
        sit under step i's backward instead of in front of layer 1.  Two batches may be in flight: a serving loop whose next
        batch is resident calls this BEFORE the forward of the current one, so the next batch's sampling runs under the whole
        current forward (a forward consumes the oldest entry that matches its batch; entries of other batches are dropped)."""
        if not (self.prefetch_sampling and points.is_cuda):
            return
        xyz = points[:, 1:4].reshape(batch_size, -1, 3)
        if len(self._prefetched) >= 2:
            self._prefetched.pop(0)
        self._prefetched.append(((points.data_ptr(), tuple(points.shape), points._version, batch_size),
                                 self._presample(xyz, points, batch_size)))

    @staticmethod
    def break_up_pc(pc):
        batch_idx = pc[:, 0]
        xyz = pc[:, 1:4].contiguous()
        features = (pc[:, 4:].contiguous() if pc.size(-1) > 4 else None)
        return batch_idx, xyz, features

    def forward(self, batch_dict):
        """batch_dict: batch_size, points (B*N, 1 + 3 + C) [bs_idx, x, y, z, ...] with the same N
        for every scene (IASSD_backbone.py:137).  Adds the reference's output keys (:188-203)."""
        with self.bn_counters():
            return self._forward(batch_dict)

    @contextlib.contextmanager
    def bn_counters(self):
        """Training: the num_batches_tracked bumps of every BatchNorm touched inside are applied as ONE _foreach_add_."""
        if not self.training or pointnet2_utils.BN_COUNTERS_PENDING is not None:
            yield
            return
        pointnet2_utils.BN_COUNTERS_PENDING = pending = []   # see pointnet2_utils.bump_bn_counter
        try:
            yield
        finally:
            pointnet2_utils.BN_COUNTERS_PENDING = None
            if pending:
                torch._foreach_add_(pending, 1)

    def _forward(self, batch_dict):
        st = self._begin(batch_dict)
        n = len(self.SA_modules)
        i0 = n
        if (self.graph_tail_infer and not self.training and not torch.is_grad_enabled() and st['encoder_xyz'][0].is_cuda
                and not torch.cuda.is_current_stream_capturing()):
            i0 = self.first_static_tail_layer()
            if i0 < 1 or any(self._layer_reads_before(i, i0) for i in range(i0, n)):
                i0 = n
        for i in range(i0):
            self._run_layer(i, st)
        if i0 < n:
            self._run_tail_graphed(i0, st)
        return self._finish(batch_dict, st)

    def _layer_reads_before(self, i, i0):
        # state index k + 1 holds the output of layer k (index 0: the raw points)
        reads = [self.layer_inputs[i]] if not isinstance(self.layer_inputs[i], list) else list(self.layer_inputs[i])
        if self.layer_types[i] == 'SA_Layer' and self.ctr_idx_list[i] != -1:
            reads.append(self.ctr_idx_list[i])
        return any(r < i0 for r in reads)

    def _run_tail_graphed(self, i0, st):
        """Layers i0.. of an inference forward as one hipGraph replay (captured on first use per input shape and per state
        of the weights).  Inputs are copied into the graph's buffers (three small tensors); the outputs stay in them."""
        from . import _lib
        n = len(self.SA_modules)
        xyz, feats, cls, bidx = st['encoder_xyz'][i0], st['encoder_features'][i0], st['li_cls_pred'], st['bidx']
        tail_params = [p for i in range(i0, n) for p in self.SA_modules[i].parameters()]
        key = (i0, st['batch_size'], tuple(xyz.shape), tuple(feats.shape), None if cls is None else tuple(cls.shape), tuple(bidx.shape),
               _lib.PARAM_EPOCH[0], _lib.WEIGHT_EPOCH[0], tuple(p._version for p in tail_params), tuple(p.data_ptr() for p in tail_params[:4]))
        if self._tail_graph is None:
            self._tail_graph = {}
        if key not in self._tail_graph:
            if len(self._tail_graph) >= 4:                  # a few input shapes in rotation at most; stale weights' graphs go first
                self._tail_graph.pop(next(iter(self._tail_graph)))
            sx, sf, sb = xyz.clone(), feats.clone(), bidx.clone()
            sc = None if cls is None else cls.clone()

            def run():
                st2 = dict(batch_size=st['batch_size'], encoder_xyz=[None] * i0 + [sx], encoder_features=[None] * i0 + [sf],
                           sa_ins_preds=[], sample_ids=[], encoder_coords=[], bidx=sb, li_cls_pred=sc, presampled={})
                for i in range(i0, n):
                    self._run_layer(i, st2)
                return st2
            side = torch.cuda.Stream(device=xyz.device)
            side.wait_stream(torch.cuda.current_stream(xyz.device))
            with torch.cuda.stream(side):
                run()                                   # warm-up outside the capture: caches (folded BatchNorm, packed planes) fill here
            torch.cuda.current_stream(xyz.device).wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                st2 = run()
            self._tail_graph[key] = (graph, (sx, sf, sc, sb), st2)
        graph, (sx, sf, sc, sb), st2 = self._tail_graph[key]
        sx.copy_(xyz); sf.copy_(feats); sb.copy_(bidx)
        if sc is not None:
            sc.copy_(cls)
        graph.replay()
        for k in ('encoder_xyz', 'encoder_features'):
            st[k] = st[k][:i0 + 1] + st2[k][i0 + 1:]
        for k in ('sa_ins_preds', 'sample_ids', 'encoder_coords'):
            st[k] = st[k] + st2[k]
        for k in ('centers', 'centers_origin', 'ctr_offsets', 'li_cls_pred'):
            if k in st2:
                st[k] = st2[k]

    # ---- the forward pass as explicit steps over a state dict (detector.IASSD replays a static tail as hipGraphs) ------
    def _begin(self, batch_dict, first_graphed=None):
        batch_size = batch_dict['batch_size']
        points = batch_dict['points']
        batch_idx, xyz, features = self.break_up_pc(points)
        assert points.shape[0] % batch_size == 0, "every scene must hold the same number of points"
        xyz = xyz.view(batch_size, -1, 3)
        features = features.view(batch_size, -1, features.shape[-1]).permute(0, 2, 1).contiguous() \
            if features is not None else None
        st = dict(batch_size=batch_size, encoder_xyz=[xyz], encoder_features=[features], sa_ins_preds=[], sample_ids=[],
                  encoder_coords=[torch.cat([batch_idx.view(batch_size, -1, 1), xyz], dim=-1)],
                  bidx=batch_idx.view(batch_size, -1), li_cls_pred=None, presampled={})
        resident = bool(batch_dict.get('inputs_resident', False))
        if self.prefetch_sampling:
            key = (points.data_ptr(), tuple(points.shape), points._version, batch_size)
            stash = None
            while self._prefetched and stash is None:
                cand = self._prefetched.pop(0)
                stash = cand if cand[0] == key else None    # an entry made for another batch is stale: dropped
            if stash is not None:
                st['presampled'] = stash[1]                 # started by prefetch() during the previous iteration
            else:
                st['presampled'] = self._presample(xyz, points if resident else None, batch_size, limit=first_graphed)
        return st

    def first_static_tail_layer(self):
        """Index of the first layer after which nothing reads a token count on the host (the layers behind the last PDA
        layer with groupers): from there to the losses every shape is static."""
        # (In dense-bf16 mode the PDA layers have no host read either, but starting the graph at layer 1 puts the D-FPS on
        # the main stream inside the graph instead of on the side stream under the previous iteration: measured 18.1 ms
        # against 15.4 ms for the KITTI iteration.)
        last = -1
        for i, m in enumerate(self.SA_modules):
            if isinstance(m, pointnet2_modules.PointnetSAModuleMSG_WithSampling_Ellipsoid) and len(m.groupers) > 0:
                last = i
        return last + 1

    def _run_layer(self, i, st):
        batch_size, bidx = st['batch_size'], st['bidx']
        xyz_input = st['encoder_xyz'][self.layer_inputs[i]]
        feature_input = st['encoder_features'][self.layer_inputs[i]]
        sample_list_id = []
        if self.layer_types[i] == 'SA_Layer':
            ctr_xyz = st['encoder_xyz'][self.ctr_idx_list[i]] if self.ctr_idx_list[i] != -1 else None
            pre = None
            if i in st['presampled']:
                ev, pidx, pxyz, pq = st['presampled'][i]
                torch.cuda.current_stream(xyz_input.device).wait_event(ev)
                if pq is not None and pq['totals'] is not None:
                    ev.synchronize()                     # the token counts in pinned memory (side stream only: no drain)
                pre = (pidx, pxyz, pq)
            li_xyz, li_features, st['li_cls_pred'], sample_list_id = self.SA_modules[i](
                xyz_input, feature_input, st['li_cls_pred'], ctr_xyz=ctr_xyz, presampled=pre)
        elif self.layer_types[i] == 'Vote_Layer':
            li_xyz, li_features, xyz_select, ctr_offsets = self.SA_modules[i](xyz_input, feature_input)
            st['centers'], st['centers_origin'], st['ctr_offsets'] = li_xyz, xyz_select, ctr_offsets
            center_origin_batch_idx = bidx[:, :xyz_select.shape[1]]
            st['encoder_coords'].append(torch.cat([center_origin_batch_idx[..., None].float(),
                                                   xyz_select.view(batch_size, -1, 3)], dim=-1))
        st['encoder_xyz'].append(li_xyz)
        li_batch_idx = bidx[:, :li_xyz.shape[1]]
        st['encoder_coords'].append(torch.cat([li_batch_idx[..., None].float(), li_xyz.view(batch_size, -1, 3)], dim=-1))
        st['encoder_features'].append(li_features)
        st['sample_ids'].append(sample_list_id)
        li_cls_pred = st['li_cls_pred']
        if li_cls_pred is not None:
            li_cls_batch_idx = bidx[:, :li_cls_pred.shape[1]]
            st['sa_ins_preds'].append(torch.cat([li_cls_batch_idx[..., None].float(),
                                                 li_cls_pred.view(batch_size, -1, li_cls_pred.shape[-1])], dim=-1))
        else:
            st['sa_ins_preds'].append([])

    def _finish(self, batch_dict, st):
        li_xyz = st['encoder_xyz'][-1]
        ctr_batch_idx = st['bidx'][:, :li_xyz.shape[1]].contiguous().view(-1)
        batch_dict['ctr_offsets'] = torch.cat((ctr_batch_idx[:, None].float(), st['ctr_offsets'].contiguous().view(-1, 3)), dim=1)
        batch_dict['centers'] = torch.cat((ctr_batch_idx[:, None].float(), st['centers'].contiguous().view(-1, 3)), dim=1)
        batch_dict['centers_origin'] = torch.cat((ctr_batch_idx[:, None].float(), st['centers_origin'].contiguous().view(-1, 3)), dim=1)
        feats = st['encoder_features'][-1]
        batch_dict['centers_features'] = feats.permute(0, 2, 1).contiguous().view(-1, feats.shape[1])
        batch_dict['ctr_batch_idx'] = ctr_batch_idx
        batch_dict['encoder_xyz'] = st['encoder_xyz']
        batch_dict['encoder_coords'] = st['encoder_coords']
        batch_dict['sa_ins_preds'] = st['sa_ins_preds']
        batch_dict['encoder_features'] = st['encoder_features']
        batch_dict['sample_list_id'] = st['sample_ids']
        return batch_dict


def build_backbone(cfg_path="once_pda_ssd.yaml", input_channels=None):
    """IASSD_Backbone from a yaml (this repo's cfgs/ or a reference model yaml)."""
    from . import config
    cfg = config.load_yaml(cfg_path)
    num_class = len(cfg.CLASS_NAMES)
    if input_channels is None:
        input_channels = cfg.get('DATA_CONFIG', {}).get('NUM_POINT_FEATURES', 4)
    return IASSD_Backbone(cfg.MODEL.BACKBONE_3D, num_class=num_class, input_channels=input_channels), cfg
