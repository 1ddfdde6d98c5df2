//! `ServerCommunication` (cubecl-runtime/src/server/base.rs `pub trait ServerCommunication`) over RCCL, through
//! `mi355_comm_init` / `mi355_all_reduce` / `mi355_send` / `mi355_recv` / `mi355_sync_collective`.
//!
//! One process, one server per device, as the reference's CUDA backend does it: the RCCL unique id of a device group
//! lives in a process-wide table keyed by `CommunicationId`, the first server to ask creates it, and every member calls
//! `mi355_comm_init` with it from its own device thread (`ncclCommInitRank` blocks until all ranks have arrived, which is
//! why `ComputeClient::ensure_init_collective` flushes right after submitting this).  Collectives run on the library's
//! per-context communication stream: each call orders that stream behind the lane's compute stream on the device, and
//! `sync_collective` orders the lane behind the communication stream -- no host wait on either side.
use crate::{
    error,
    ffi::*,
    server::{Mi355Server, Pending},
};
use cubecl_common::device::DeviceId;
use cubecl_environment::{backtrace::BackTrace, stream::StreamId};
use cubecl_ir::{ElemType, FloatKind, IntKind, UIntKind};
use cubecl_runtime::server::{
    BufferBinding, CommunicationId, ComputeServer, CopyDescriptor, Handle, ReduceOperation, ServerCommunication, ServerError,
};
use std::{
    collections::{HashMap, hash_map::Entry},
    sync::{Mutex, OnceLock},
};

static UNIQUE_IDS: OnceLock<Mutex<HashMap<CommunicationId, [u8; MI355_UNIQUE_ID_BYTES]>>> = OnceLock::new();

fn unique_id(group: &CommunicationId) -> Result<[u8; MI355_UNIQUE_ID_BYTES], ServerError> {
    let mut table = UNIQUE_IDS.get_or_init(Default::default).lock().unwrap();
    if let Some(id) = table.get(group) {
        return Ok(*id);
    }
    let mut id = [0u8; MI355_UNIQUE_ID_BYTES];
    error::check(core::ptr::null_mut(), unsafe { mi355_comm_unique_id(id.as_mut_ptr()) })?;
    table.insert(group.clone(), id);
    Ok(id)
}

/// `MI355_DTYPE_*` for what RCCL can reduce, with the element count of a buffer of `bytes` bytes.
fn wire_type(dtype: ElemType, bytes: u64) -> Result<(i32, u64), ServerError> {
    let (code, width) = match dtype {
        ElemType::Float(FloatKind::F32) | ElemType::Float(FloatKind::Flex32) => (MI355_DTYPE_F32, 4),
        ElemType::Float(FloatKind::F64) => (MI355_DTYPE_F64, 8),
        ElemType::Float(FloatKind::F16) => (MI355_DTYPE_F16, 2),
        ElemType::Float(FloatKind::BF16) => (MI355_DTYPE_BF16, 2),
        ElemType::Int(IntKind::I32) => (MI355_DTYPE_I32, 4),
        ElemType::Int(IntKind::I64) => (MI355_DTYPE_I64, 8),
        ElemType::Int(IntKind::I8) => (MI355_DTYPE_I8, 1),
        ElemType::UInt(UIntKind::U32) => (MI355_DTYPE_U32, 4),
        ElemType::UInt(UIntKind::U64) => (MI355_DTYPE_U64, 8),
        ElemType::UInt(UIntKind::U8) | ElemType::Bool => (MI355_DTYPE_U8, 1),
        other => {
            return Err(ServerError::Generic {
                reason: format!("{other:?} has no RCCL wire type"),
                backtrace: BackTrace::capture(),
            });
        }
    };
    Ok((code, bytes / width))
}

fn group_of(mut device_ids: Vec<DeviceId>) -> (CommunicationId, Vec<DeviceId>) {
    device_ids.sort();
    (CommunicationId::from(device_ids.clone()), device_ids)
}

impl Mi355Server {
    fn communicator(&self, device_ids: Vec<DeviceId>) -> Result<(*mut mi355_comm, Vec<DeviceId>), ServerError> {
        let (group, sorted) = group_of(device_ids);
        match self.communicators.get(&group) {
            Some(comm) => Ok((*comm, sorted)),
            None => Err(ServerError::Generic {
                reason: "collective on a device group that was never given to comm_init".into(),
                backtrace: BackTrace::capture(),
            }),
        }
    }

    fn rank_in(&self, sorted: &[DeviceId], device: DeviceId) -> Result<i32, ServerError> {
        sorted.iter().position(|id| id.index_id == device.index_id).map(|p| p as i32).ok_or_else(|| ServerError::Generic {
            reason: format!("{device} is not a member of the device group"),
            backtrace: BackTrace::capture(),
        })
    }
}

impl Mi355Server {
    /// The exchange step of a sharded fused sum + arg-max (config C4 of the benchmark) without a host round trip, beyond what
    /// `ServerCommunication` offers (the trait has no all-gather): `record` is the 16-byte `{f32 max, f32 partial sum, u64 local
    /// index}` the local pass wrote (`mi355_sum_argmax_f32` with `out_val = rec`, `out_sum = rec + 4`, `out_idx = rec + 8`),
    /// `gathered` holds one record per rank.  ONE library call (`mi355_sum_argmax_exchange`, ABI 9): one collective
    /// (`mi355_all_gather`), the comm -> compute fence, then a 64-lane kernel that adds the partial sums in rank order and folds the
    /// candidates with the single-GPU rule: the same bits on every rank.  `index_base[r]` = global index of rank r's first element --
    /// a HOST slice: the library reads it on the host while it queues the combine kernel (`include/mi355cube.h`, "HOST array"; until
    /// round 6 this method took a device binding and handed its pointer over, which the first call would have dereferenced on the
    /// host).  The Python mirror is `cubecl_amd/runtime.py::ComputeClient::sum_argmax_exchange` (`sharded.py`, mode "gather"); the
    /// reference's own shape (all_reduce of the sum + a gather of the candidates) needs nothing beyond the trait.
    #[allow(clippy::too_many_arguments)]
    pub fn sum_argmax_exchange(&mut self, record: BufferBinding, gathered: BufferBinding, index_base: &[u64], out_sum: BufferBinding,
                               out_value: BufferBinding, out_index: BufferBinding, stream_id: StreamId, device_ids: Vec<DeviceId>)
                               -> Result<(), ServerError> {
        let ctx = self.ctx;
        let (comm, sorted) = self.communicator(device_ids)?;
        let ranks = sorted.len() as u64;
        let mut pass = self.pass(stream_id, [&record, &gathered, &out_sum, &out_value, &out_index].into_iter(), Pending::Surface)?;
        let (rec, all) = (pass.slice(record)?, pass.slice(gathered)?);
        let (sum, val, idx) = (pass.slice(out_sum)?, pass.slice(out_value)?, pass.slice(out_index)?);
        if rec.size < 16 || all.size < 16 * ranks || index_base.len() as u64 != ranks {
            return Err(ServerError::Generic { reason: "sum_argmax_exchange: a 16-byte record per rank and one u64 base per rank".into(),
                                              backtrace: BackTrace::capture() });
        }
        error::check(ctx, unsafe {
            mi355_sum_argmax_exchange(ctx, comm, pass.sys(), rec.ptr, all.ptr, index_base.as_ptr(), sum.ptr as *mut f32, val.ptr as *mut f32,
                                      idx.ptr as *mut u64)
        })
    }
}

pub(crate) fn destroy_all(server: &mut Mi355Server) {
    for (_, comm) in server.communicators.drain() {
        unsafe { mi355_comm_destroy(server.ctx, comm) };
    }
}

impl ServerCommunication for Mi355Server {
    const SERVER_COMM_ENABLED: bool = true;

    fn sync_collective(&mut self, stream_id: StreamId) -> Result<(), ServerError> {
        let ctx = self.ctx;
        let mut pass = self.pass_alone(stream_id, Pending::Keep)?;
        error::check(ctx, unsafe { mi355_sync_collective(ctx, pass.sys()) })
    }

    fn comm_init(&mut self, device_ids: Vec<DeviceId>) -> Result<(), ServerError> {
        let (group, sorted) = group_of(device_ids);
        let rank = self.rank_in(&sorted, self.device_id)?;
        if let Entry::Vacant(slot) = self.communicators.entry(group.clone()) {
            let id = unique_id(&group)?;
            let mut comm: *mut mi355_comm = core::ptr::null_mut();
            error::check(self.ctx, unsafe { mi355_comm_init(self.ctx, id.as_ptr(), rank, sorted.len() as i32, &mut comm) })?;
            slot.insert(comm);
            self.utilities.initialized_comms.write().insert(group);
        }
        Ok(())
    }

    fn all_reduce(&mut self, src: BufferBinding, dst: BufferBinding, dtype: ElemType, stream_id: StreamId, op: ReduceOperation,
                  device_ids: Vec<DeviceId>) -> Result<(), ServerError> {
        let ctx = self.ctx;
        let (comm, _) = self.communicator(device_ids)?;
        let mut pass = self.pass(stream_id, [&src, &dst].into_iter(), Pending::Surface)?;
        let input = pass.slice(src)?;
        let output = pass.slice(dst)?;
        let (wire, count) = wire_type(dtype, input.size)?;
        let op = match op {
            ReduceOperation::Sum => MI355_REDUCE_SUM,
            ReduceOperation::Mean => MI355_REDUCE_MEAN,
        };
        error::check(ctx, unsafe { mi355_all_reduce(ctx, comm, pass.sys(), input.ptr, output.ptr, count, wire, op) })
    }

    fn send(&mut self, desc: CopyDescriptor, dtype: ElemType, stream_id: StreamId, device_id_dst: DeviceId) -> Result<(), ServerError> {
        let ctx = self.ctx;
        let (comm, sorted) = self.communicator(vec![device_id_dst, self.device_id])?;
        let peer = self.rank_in(&sorted, device_id_dst)?;
        let mut pass = self.pass(stream_id, [&desc.handle].into_iter(), Pending::Keep)?;
        let source = pass.slice(desc.handle)?;
        let (wire, count) = wire_type(dtype, source.size)?;
        error::check(ctx, unsafe { mi355_send(ctx, comm, pass.sys(), source.ptr, count, wire, peer) })
    }

    fn recv(&mut self, handle: Handle, dtype: ElemType, stream_id: StreamId, device_id_src: DeviceId) -> Result<(), ServerError> {
        let ctx = self.ctx;
        let (comm, sorted) = self.communicator(vec![device_id_src, self.device_id])?;
        let peer = self.rank_in(&sorted, device_id_src)?;
        // the client made `handle` up (to_client_tensor): give it memory on this device first
        self.initialize_memory(handle.memory.clone(), handle.size(), stream_id);
        let binding = handle.binding();
        let mut pass = self.pass(stream_id, [&binding].into_iter(), Pending::Keep)?;
        let target = pass.slice(binding)?;
        let (wire, count) = wire_type(dtype, target.size)?;
        error::check(ctx, unsafe { mi355_recv(ctx, comm, pass.sys(), target.ptr, count, wire, peer) })
    }
}
