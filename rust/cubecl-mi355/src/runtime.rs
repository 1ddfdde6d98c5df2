//! `Mi355Runtime`: the `Runtime` implementation (crates/cubecl-runtime/src/runtime.rs:14-52) and the
//! gfx950 property block (reference analogue crates/cubecl-hip/src/runtime.rs:155-220).
use crate::{ffi::*, server::Mi355Server, storage::Mi355Storage};
use cubecl_common::device::{Device, DeviceId};
use cubecl_ir::{features::*, DeviceProperties, HardwareProperties, MemoryDeviceProperties, TargetProperties};
use cubecl_runtime::{client::ComputeClient, memory_management::MemoryManagement, runtime::Runtime, server::*};
use cubecl_zspace::{Shape, Strides};
use std::sync::{Arc, Mutex, OnceLock};

#[derive(Debug, Clone, Default, PartialEq, Eq, Hash)]
pub struct Mi355Device { pub index: usize }

impl Device for Mi355Device {
    fn from_id(id: DeviceId) -> Self { Self { index: id.index_id as usize } }
    fn to_id(&self) -> DeviceId { DeviceId { type_id: 0, index_id: self.index as u32 } }
}

/// Kernels are built ahead of time with hipcc; this "compiler" only carries the representation type.
#[derive(Clone, Debug, Default)]
pub struct AotCompiler;
// impl cubecl_runtime::compiler::Compiler for AotCompiler { type Representation = crate::server::ExternalKernel; ... }
// (compile() of an external CubeTask returns the caller's code object unchanged; elided: pure plumbing)

#[derive(Debug, Clone)]
pub struct Mi355Runtime;

impl Runtime for Mi355Runtime {
    type Compiler = AotCompiler;
    type Server = Mi355Server;
    type Device = Mi355Device;

    fn client(device: &Self::Device) -> ComputeClient<Self> { ComputeClient::load(device) }
    fn name(_client: &ComputeClient<Self>) -> &'static str { "mi355" }
    fn require_array_lengths() -> bool { true }                       // as the HIP backend (runtime.rs:267-269)
    fn max_cube_count() -> (u32, u32, u32) { (i32::MAX as u32, 65535, 65535) }
    fn can_read_tensor(shape: &Shape, strides: &Strides) -> bool {
        cubecl_zspace::striding::has_pitched_row_major_strides(shape, strides)
    }
    fn target_properties() -> TargetProperties {
        // wave64 MFMA register layouts; only consumed by manual-MMA kernels
        TargetProperties { mma: cubecl_ir::MmaProperties { const_plane_size: 64, ..Default::default() } }
    }
    fn enumerate_devices(_type_id: u16, _info: &()) -> Vec<DeviceId> {
        let mut n = 0i32;
        unsafe { mi355_device_count(&mut n) };
        (0..n.max(0) as u32).map(|i| DeviceId { type_id: 0, index_id: i }).collect()
    }
}

/// Turns the C property block into the structs cubek and the benches query.
pub(crate) fn build_memory_and_utilities(p: &mi355_device_props_t, storage: Mi355Storage, device: DeviceId)
    -> (MemoryManagement<Mi355Storage>, Arc<ServerUtilities<Mi355Server>>) {
    let hardware = HardwareProperties {
        load_width: p.load_width_bits,                               // 128
        plane_size_min: p.plane_size_min,                            // 64
        plane_size_max: p.plane_size_max,                            // 64
        max_bindings: p.max_bindings,
        max_shared_memory_size: p.max_shared_memory_size as usize,
        max_cube_count: (p.max_cube_count[0], p.max_cube_count[1], p.max_cube_count[2]),
        max_units_per_cube: p.max_units_per_cube,
        max_cube_dim: (p.max_cube_dim[0], p.max_cube_dim[1], p.max_cube_dim[2]),
        num_streaming_multiprocessors: Some(p.num_streaming_multiprocessors),   // 256; the reference leaves None
        num_tensor_cores: Some(p.num_tensor_cores),
        min_tensor_cores_dim: Some(p.min_tensor_cores_dim),
        num_cpu_cores: None,
        max_vector_size: cubecl_ir::VectorSize::MAX,
    };
    let memory_props = MemoryDeviceProperties { max_page_size: p.max_page_size, alignment: p.mem_alignment };
    let mut props = DeviceProperties::new(Default::default(), memory_props.clone(), hardware,
                                          cubecl_common::profile::TimingMethod::Device);
    props.features.plane.insert(Plane::Ops);
    props.features.plane.insert(Plane::NonUniformControlFlow);
    for c in &p.mma_configs[..p.num_mma_configs as usize] {
        // bf16/f16 32x32x16 + 16x16x32, f32 32x32x2 + 16x16x4: makes testgen_cmma! run instead of skip
        props.features.matmul.cmma.insert(MmaConfig { a_type: elem(c.a_type), b_type: elem(c.b_type),
                                                      cd_type: elem(c.cd_type), m: c.m, n: c.n, k: c.k });
    }
    let memory = MemoryManagement::from_configuration(storage, &memory_props, Default::default());
    let utilities = Arc::new(ServerUtilities::new(props, Default::default(), (),
                                                  cubecl_runtime::allocator::PitchedMemoryLayoutPolicy::new(p.mem_alignment as usize)));
    let _ = device;
    (memory, utilities)
}

fn elem(code: i32) -> cubecl_ir::StorageType {
    use cubecl_ir::{ElemType, FloatKind};
    match code {
        MI355_DTYPE_BF16 => ElemType::Float(FloatKind::BF16).into(),
        MI355_DTYPE_F16 => ElemType::Float(FloatKind::F16).into(),
        _ => ElemType::Float(FloatKind::F32).into(),
    }
}

/// One RCCL unique id per device set, shared by the per-device server threads of this process.
pub(crate) fn unique_id_for(id: &CommunicationId) -> [u8; MI355_UNIQUE_ID_BYTES] {
    static IDS: OnceLock<Mutex<std::collections::HashMap<u64, [u8; MI355_UNIQUE_ID_BYTES]>>> = OnceLock::new();
    let mut map = IDS.get_or_init(Default::default).lock().unwrap();
    *map.entry(id.id).or_insert_with(|| {
        let mut uid = [0u8; MI355_UNIQUE_ID_BYTES];
        unsafe { mi355_comm_unique_id(uid.as_mut_ptr()) };
        uid
    })
}
