//! `Mi355Runtime`: `Runtime` (cubecl-runtime/src/runtime.rs), `Mi355Device`: `Device`, and `DeviceService::init` for
//! [`Mi355Server`] -- where the device is opened (`mi355_ctx_create`) and described.  Everything the reference's HIP
//! backend learns from `hipGetDeviceProperties` plus an architecture table comes here from `mi355_device_props`, one
//! struct the library fills for gfx950: limits, the supported-type and atomic registry, the MFMA shapes, and the lane
//! layout of an MFMA fragment.
use crate::{error, ffi::*, server::Mi355Server};
use cubecl_common::{
    device::{Device, DeviceId, DeviceService, ServerUtilitiesHandle},
    profile::TimingMethod,
};
use cubecl_cpp::shared::{CompilationOptions, CppSupportedFeatures};
use cubecl_ir::{
    AddressType, ContiguousElements, DeviceIdentity, DeviceProperties, ElemType, FloatKind, HardwareProperties, IntKind,
    MemoryDeviceProperties, MmaProperties, TargetProperties, Type, UIntKind, VectorSize,
    features::{AtomicUsage, EnumSet, MmaConfig, Plane, ScaledMmaConfig, TypeUsage},
    interfaces::{TypeExt, TypedExt},
    pliron::{context::Context, r#type::TypedHandle},
    types::matrix::{MatrixIdent, MatrixLayout, MatrixType},
};
use cubecl_runtime::{
    allocator::PitchedMemoryLayoutPolicy,
    client::ComputeClient,
    logging::ServerLogger,
    memory_management::MemoryConfiguration,
    runtime::Runtime,
    server::{ComputeServer, ServerUtilities},
};
use cubecl_zspace::{Shape, Strides, striding::has_pitched_row_major_strides};
use std::{ffi::CStr, sync::Arc};

/// A MI355X by its HIP ordinal.
#[derive(Clone, PartialEq, Eq, Default, Hash)]
pub struct Mi355Device {
    pub index: usize,
}

impl Mi355Device {
    pub fn new(index: usize) -> Self {
        Self { index }
    }
}

impl core::fmt::Debug for Mi355Device {
    fn fmt(&self, f: &mut core::fmt::Formatter<'_>) -> core::fmt::Result {
        write!(f, "Mi355Device({})", self.index)
    }
}

impl Device for Mi355Device {
    fn from_id(device_id: DeviceId) -> Self {
        Self { index: device_id.index_id as usize }
    }

    fn to_id(&self) -> DeviceId {
        DeviceId { type_id: 0, index_id: self.index as u16 }
    }
}

/// `MI355_DTYPE_*` -> the reference's element type (`None` for codes that have no `ElemType`).
pub(crate) fn elem_of(code: i32) -> Option<ElemType> {
    Some(match code {
        MI355_DTYPE_F32 => ElemType::Float(FloatKind::F32),
        MI355_DTYPE_BF16 => ElemType::Float(FloatKind::BF16),
        MI355_DTYPE_F16 => ElemType::Float(FloatKind::F16),
        MI355_DTYPE_F64 => ElemType::Float(FloatKind::F64),
        MI355_DTYPE_FLEX32 => ElemType::Float(FloatKind::Flex32),
        MI355_DTYPE_F8E4M3 => ElemType::Float(FloatKind::E4M3),
        MI355_DTYPE_F8E5M2 => ElemType::Float(FloatKind::E5M2),
        MI355_DTYPE_F4E2M1X2 => ElemType::Float(FloatKind::E2M1x2),
        MI355_DTYPE_UE8M0 => ElemType::Float(FloatKind::UE8M0),
        MI355_DTYPE_I8 => ElemType::Int(IntKind::I8),
        MI355_DTYPE_I16 => ElemType::Int(IntKind::I16),
        MI355_DTYPE_I32 => ElemType::Int(IntKind::I32),
        MI355_DTYPE_I64 => ElemType::Int(IntKind::I64),
        MI355_DTYPE_U8 => ElemType::UInt(UIntKind::U8),
        MI355_DTYPE_U16 => ElemType::UInt(UIntKind::U16),
        MI355_DTYPE_U32 => ElemType::UInt(UIntKind::U32),
        MI355_DTYPE_U64 => ElemType::UInt(UIntKind::U64),
        MI355_DTYPE_BOOL => ElemType::Bool,
        MI355_DTYPE_INDEX => ElemType::Index,
        _ => return None,
    })
}

fn layout_of(code: u32) -> MatrixLayout {
    match code {
        MI355_LAYOUT_ROW_MAJOR => MatrixLayout::RowMajor,
        MI355_LAYOUT_COL_MAJOR => MatrixLayout::ColMajor,
        _ => MatrixLayout::Undefined,
    }
}

/// Elements of one MFMA fragment a lane holds contiguously: 128 bits of consecutive `k` for A and B (8 bf16/f16, 16 fp8),
/// four consecutive rows of one column for the f32 accumulator -- the `v_mfma_f32_32x32x16_bf16` register layout the
/// kernels in cubecl_amd/csrc are written against (`mi355_mma_properties`).
fn mfma_contiguous_elements(ctx: &Context, ident: MatrixIdent, matrix: TypedHandle<MatrixType>) -> VectorSize {
    let matrix = matrix.deref(ctx);
    match ident {
        MatrixIdent::A | MatrixIdent::B => 16 / matrix.elem_ty.size(ctx),
        MatrixIdent::Accumulator => 4,
    }
}

fn text(raw: &[core::ffi::c_char]) -> String {
    unsafe { CStr::from_ptr(raw.as_ptr()) }.to_string_lossy().into_owned()
}

/// `mi355_device_props_t` -> `DeviceProperties`, with the registries the reference's backends fill by hand
/// (cubecl-cpp/src/shared/base.rs `register_supported_types`, shared/mma.rs `register_mma_features`).
fn describe(props: &mi355_device_props_t) -> (DeviceProperties, MemoryDeviceProperties) {
    let memory = MemoryDeviceProperties { max_page_size: props.max_page_size, alignment: props.mem_alignment };
    let hardware = HardwareProperties {
        load_width: props.load_width_bits,
        plane_size_min: props.plane_size_min,
        plane_size_max: props.plane_size_max,
        max_bindings: props.max_bindings,
        max_shared_memory_size: props.max_shared_memory_size as usize,
        max_cube_count: (props.max_cube_count[0], props.max_cube_count[1], props.max_cube_count[2]),
        max_units_per_cube: props.max_units_per_cube,
        max_cube_dim: (props.max_cube_dim[0], props.max_cube_dim[1], props.max_cube_dim[2]),
        num_streaming_multiprocessors: Some(props.num_streaming_multiprocessors),
        num_cpu_cores: None,
        num_tensor_cores: Some(props.num_tensor_cores),
        min_tensor_cores_dim: Some(props.min_tensor_cores_dim),
        max_vector_size: VectorSize::MAX,
        cube_mma_reserved_shared_memory: 0,
    };
    let mut out = DeviceProperties::new(
        Default::default(),
        memory.clone(),
        hardware,
        TimingMethod::System,
        DeviceIdentity { name: text(&props.name), fingerprint: text(&props.fingerprint) },
    );

    if props.address_types & MI355_ADDRESS_TYPE_U32 != 0 {
        out.register_address_type(AddressType::U32);
    }
    if props.address_types & MI355_ADDRESS_TYPE_U64 != 0 {
        out.register_address_type(AddressType::U64);
    }
    for entry in &props.type_usage[..props.num_type_usage as usize] {
        let Some(elem) = elem_of(entry.dtype) else { continue };
        let mut usage = EnumSet::<TypeUsage>::empty();
        for (bit, flag) in [(MI355_TYPE_USAGE_CONVERSION, TypeUsage::Conversion), (MI355_TYPE_USAGE_ARITHMETIC, TypeUsage::Arithmetic),
                            (MI355_TYPE_USAGE_DOT_PRODUCT, TypeUsage::DotProduct), (MI355_TYPE_USAGE_BUFFER, TypeUsage::Buffer)] {
            if entry.usage & bit != 0 {
                usage |= flag;
            }
        }
        out.register_type_usage(elem, usage);
    }
    for entry in &props.atomic_usage[..props.num_atomic_usage as usize] {
        let Some(elem) = elem_of(entry.dtype) else { continue };
        let mut usage = EnumSet::<AtomicUsage>::empty();
        for (bit, flag) in [(MI355_ATOMIC_LOAD_STORE, AtomicUsage::LoadStore), (MI355_ATOMIC_EXCHANGE, AtomicUsage::Exchange), (MI355_ATOMIC_ADD, AtomicUsage::Add),
                            (MI355_ATOMIC_MIN_MAX, AtomicUsage::MinMax), (MI355_ATOMIC_BITWISE, AtomicUsage::Bitwise),
                            (MI355_ATOMIC_COMPARE_EXCHANGE, AtomicUsage::CompareExchange)] {
            if entry.usage & bit != 0 {
                usage |= flag;
            }
        }
        out.register_atomic_type_usage(Type::atomic(elem), usage);
    }

    out.features.memory_reinterpret = true;
    out.features.alignment = true;
    if props.plane_ops != 0 {
        out.features.plane.insert(Plane::Ops);
    }
    if props.plane_non_uniform != 0 {
        out.features.plane.insert(Plane::NonUniformControlFlow);
    }
    for mma in &props.mma_configs[..props.num_mma_configs as usize] {
        if let (Some(a_type), Some(b_type), Some(cd_type)) = (elem_of(mma.a_type), elem_of(mma.b_type), elem_of(mma.cd_type)) {
            out.features.matmul.mma.insert(MmaConfig { a_type, b_type, cd_type, m: mma.m, n: mma.n, k: mma.k });
        }
    }
    for mma in &props.scaled_mma_configs[..props.num_scaled_mma_configs as usize] {
        if let (Some(a_type), Some(b_type), Some(cd_type), Some(scales_type)) =
            (elem_of(mma.a_type), elem_of(mma.b_type), elem_of(mma.cd_type), elem_of(mma.scales_type))
        {
            out.features.matmul.scaled_mma.insert(ScaledMmaConfig { a_type, b_type, cd_type, scales_type, m: mma.m, n: mma.n, k: mma.k,
                                                                    scales_factor: mma.scales_factor });
        }
    }
    (out, memory)
}

impl DeviceService for Mi355Server {
    fn init(device_id: DeviceId) -> Self {
        let device = Mi355Device::from_id(device_id);
        let mut ctx: *mut mi355_ctx = core::ptr::null_mut();
        let rc = unsafe { mi355_ctx_create(device.index as i32, &mut ctx) };
        assert_eq!(rc, MI355_OK, "mi355_ctx_create({}): {}", device.index, error::last_message(core::ptr::null_mut()));

        let mut raw = core::mem::MaybeUninit::<mi355_device_props_t>::zeroed();
        let rc = unsafe { mi355_device_props(ctx, raw.as_mut_ptr()) };
        assert_eq!(rc, MI355_OK, "mi355_device_props: {}", error::last_message(ctx));
        let raw = unsafe { raw.assume_init() };
        assert_eq!(raw.abi_version, MI355_ABI_VERSION, "libmi355cube.so and this crate were built from different headers");

        let (properties, memory) = describe(&raw);
        let options = CompilationOptions {
            warp_size: raw.plane_size_max as usize,
            supports_features: CppSupportedFeatures { fast_math: true, ..Default::default() },
            amd_wmma: None, // CDNA has MFMA, not WMMA
        };
        let logger = Arc::new(ServerLogger::default());
        let policy = PitchedMemoryLayoutPolicy::new(properties.memory.alignment as usize);
        let utilities = ServerUtilities::new(properties, logger, (), policy);
        Mi355Server::new(ctx, device_id, memory, MemoryConfiguration::default(), options, utilities)
    }

    fn utilities(&self) -> ServerUtilitiesHandle {
        ComputeServer::utilities(self) as ServerUtilitiesHandle
    }
}

#[derive(Debug, Clone)]
pub struct Mi355Runtime;

impl Runtime for Mi355Runtime {
    type Compiler = crate::compiler::Mi355Compiler;
    type Server = Mi355Server;
    type Device = Mi355Device;

    fn client(device: &Self::Device) -> ComputeClient<Self> {
        ComputeClient::load(device)
    }

    fn name(_client: &ComputeClient<Self>) -> &'static str {
        "mi355"
    }

    fn require_array_lengths() -> bool {
        true
    }

    fn max_cube_count() -> (u32, u32, u32) {
        (i32::MAX as u32, u16::MAX as u32, u16::MAX as u32)
    }

    fn can_read_tensor(shape: &Shape, strides: &Strides) -> bool {
        shape.is_empty() || has_pitched_row_major_strides(shape, strides)
    }

    /// Static by the trait's signature, so these are the gfx950 constants `mi355_mma_properties` reports at run time
    /// (tests/test_client_cpu.py holds the two together).
    fn target_properties() -> TargetProperties {
        TargetProperties {
            mma: MmaProperties {
                register_size_bits: 32,
                const_plane_size: 64,
                register_layout_a: layout_of(MI355_LAYOUT_ROW_MAJOR),
                register_layout_b: layout_of(MI355_LAYOUT_COL_MAJOR),
                register_layout_acc: layout_of(MI355_LAYOUT_COL_MAJOR),
                register_duplication_a: 1,
                register_duplication_b: 1,
                register_duplication_acc: 1,
                contiguous_elements: ContiguousElements::new(mfma_contiguous_elements),
            },
        }
    }

    fn enumerate_devices(_type_id: u16, _info: &<Self::Server as ComputeServer>::Info) -> Vec<DeviceId> {
        let mut count = 0i32;
        match unsafe { mi355_device_count(&mut count) } {
            MI355_OK => (0..count.max(0) as u16).map(|index_id| DeviceId { type_id: 0, index_id }).collect(),
            _ => Vec::new(),
        }
    }
}
