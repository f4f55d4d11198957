// UNVERIFIED (no Rust toolchain in the build image).  src/raytracer/hip.rs: bindings of include/trayracer_hip.h and a
// safe wrapper.  Every unsafe block calls one C entry point with pointers that outlive the call (the library borrows
// its inputs for the duration of the call only).
#![allow(unsafe_code)]

use crate::scene::{Object, Scene, Skybox};
use anyhow::{anyhow, bail};
use std::ffi::CStr;
use std::os::raw::{c_char, c_int, c_void};
use std::sync::atomic::{AtomicBool, AtomicI32, Ordering};

#[repr(C)]
#[derive(Clone, Copy, Default)]
struct TrrMaterial { kd: [f32; 3], ks: [f32; 3], ns: f32, dissolve: f32, illum: i32, texture: i32, flags: u32 }
#[repr(C)]
#[derive(Clone, Copy)]
struct TrrImage { rgb: *const u8, width: u32, height: u32 }
#[repr(C)]
struct TrrObject {
    triangles: *const f32, material_index: *const i32, n_triangles: u32,
    materials: *const TrrMaterial, n_materials: u32, transform: [f32; 16], inverse: [f32; 16],
}
#[repr(C)]
struct TrrLight { position: [f32; 3], color: [f32; 3], intensity: f32 }
#[repr(C)]
struct TrrSettings {
    ambient_color: [f32; 3], ambient_intensity: f32, skybox_is_image: i32, skybox_color: [f32; 3],
    skybox_image: TrrImage, anti_aliasing: i32, samples: u32, seed: u64,
}
#[repr(C)]
struct TrrCamera { position: [f32; 3], look_at: [f32; 3], up: [f32; 3], fov: f32, width: u32, height: u32 }

type ProgressCb = extern "C" fn(*mut c_void, u32, u32);
type BlockCb = extern "C" fn(*mut c_void, u32, u32, u32, u32, *const u8);

extern "C" {
    fn trr_create(device: c_int, out: *mut *mut c_void) -> c_int;
    fn trr_create_multi(device_ids: *const c_int, n_devices: c_int, out: *mut *mut c_void) -> c_int;
    fn trr_destroy(ctx: *mut c_void);
    fn trr_last_error(ctx: *const c_void) -> *const c_char;
    fn trr_scene_create_ex(ctx: *mut c_void, objs: *const TrrObject, n: u32, tex: *const TrrImage, n_tex: u32, build: u32,
                           out: *mut *mut c_void) -> c_int;
    fn trr_scene_destroy(scene: *mut c_void);
    fn trr_scene_set_lights(scene: *mut c_void, l: *const TrrLight, n: u32) -> c_int;
    fn trr_scene_set_settings(scene: *mut c_void, s: *const TrrSettings) -> c_int;
    fn trr_scene_set_camera(scene: *mut c_void, c: *const TrrCamera) -> c_int;
    fn trr_render_blocks(ctx: *mut c_void, scene: *mut c_void, delta: f32, max_depth: u32, flags: u32, blocks_per_batch: u32,
                         cancel: *const i32, sink: Option<BlockCb>, progress: Option<ProgressCb>, user: *mut c_void,
                         stats: *mut c_void) -> c_int;
}

pub const BUILD_HOST: u32 = 0;    // TRR_BUILD_HOST: binned-SAH tree on the CPU cores (final renders)
pub const BUILD_DEVICE: u32 = 1;  // TRR_BUILD_DEVICE: LBVH on the GPU (while the scene is being edited)
const ERR_CANCELLED: c_int = -4;

/// One GPU of this process, or all GPUs of the node behind one context (`trr_ctx`).  Not `Sync`: exactly one render is in
/// flight (ui/statusbar.rs:142-157).
pub struct Backend { ctx: *mut c_void }
unsafe impl Send for Backend {}

impl Backend {
    pub fn new(device: i32) -> anyhow::Result<Self> {
        let mut ctx = std::ptr::null_mut();
        let rc = unsafe { trr_create(device, &mut ctx) };
        if rc != 0 { bail!("no HIP device: {}", last_error(std::ptr::null())); }
        Ok(Self { ctx })
    }

    /// Several GPUs behind ONE context: the library replicates every uploaded scene, deals the 20x20 block grid round-robin
    /// (block id % N), and gathers the strips with one ncclGather; `render_blocks` below needs no change to use them.
    pub fn new_multi(devices: &[i32]) -> anyhow::Result<Self> {
        let mut ctx = std::ptr::null_mut();
        let rc = unsafe { trr_create_multi(devices.as_ptr(), devices.len() as c_int, &mut ctx) };
        if rc != 0 { bail!("trr_create_multi: {}", last_error(std::ptr::null())); }
        Ok(Self { ctx })
    }

    /// Scene::clone + Bvh::build (render.rs:60, object.rs:156) become one upload.
    pub fn upload(&self, scene: &Scene, build: u32) -> anyhow::Result<GpuScene<'_>> {
        let mut tris: Vec<Vec<f32>> = Vec::new();
        let mut mat_idx: Vec<Vec<i32>> = Vec::new();
        let mut mats: Vec<Vec<TrrMaterial>> = Vec::new();
        let mut textures: Vec<TrrImage> = Vec::new();
        for o in &scene.objects {
            tris.push(flatten_triangles(o));
            mat_idx.push(o.triangles.iter().map(|t| t.material_index.map_or(-1, |i| i as i32)).collect());
            mats.push(o.materials.iter().map(|m| {
                let mut d = TrrMaterial { texture: -1, illum: m.illumination_model.as_i32(), ..TrrMaterial::default() };
                if let Some(c) = m.diffuse_color { d.kd = [c.x, c.y, c.z]; d.flags |= 1; }
                if let Some(c) = m.specular_color { d.ks = [c.x, c.y, c.z]; d.flags |= 2; }
                if let Some(n) = m.specular_exponent { d.ns = n; d.flags |= 4; }
                if let Some(x) = m.dissolve { d.dissolve = x; d.flags |= 8; }
                if let Some(img) = &m.diffuse_texture {
                    d.texture = textures.len() as i32;
                    textures.push(TrrImage { rgb: img.as_raw().as_ptr(), width: img.width(), height: img.height() });
                }
                d
            }).collect());
        }
        let objs: Vec<TrrObject> = scene.objects.iter().enumerate().map(|(i, o)| {
            // the host's own nalgebra computes both matrices, so the device inherits its bits (object.rs:177-182)
            let fwd = o.transform().to_homogeneous();
            let inv = o.transform().inverse().to_homogeneous();
            let mut t = TrrObject {
                triangles: tris[i].as_ptr(), material_index: mat_idx[i].as_ptr(), n_triangles: o.triangles.len() as u32,
                materials: mats[i].as_ptr(), n_materials: mats[i].len() as u32, transform: [0.0; 16], inverse: [0.0; 16],
            };
            t.transform.copy_from_slice(fwd.as_slice());
            t.inverse.copy_from_slice(inv.as_slice());
            t
        }).collect();
        let mut handle = std::ptr::null_mut();
        self.check(unsafe { trr_scene_create_ex(self.ctx, objs.as_ptr(), objs.len() as u32, textures.as_ptr(), textures.len() as u32,
                                                build, &mut handle) })?;
        let gpu = GpuScene { backend: self, handle };

        let lights: Vec<TrrLight> = scene.lights.iter().map(|l| TrrLight {
            position: [l.position.x, l.position.y, l.position.z], color: [l.color.x, l.color.y, l.color.z], intensity: l.intensity,
        }).collect();
        self.check(unsafe { trr_scene_set_lights(handle, lights.as_ptr(), lights.len() as u32) })?;

        let s = &scene.settings;
        let mut st = TrrSettings {
            ambient_color: [s.ambient_color.x, s.ambient_color.y, s.ambient_color.z], ambient_intensity: s.ambient_intensity,
            skybox_is_image: 0, skybox_color: [0.0; 3], skybox_image: TrrImage { rgb: std::ptr::null(), width: 0, height: 0 },
            anti_aliasing: i32::from(s.anti_aliasing), samples: s.samples, seed: 1,
        };
        match &s.skybox {
            Skybox::Color(c) => st.skybox_color = [c.x, c.y, c.z],
            Skybox::Image { image, .. } => {
                st.skybox_is_image = 1;
                st.skybox_image = TrrImage { rgb: image.as_raw().as_ptr(), width: image.width(), height: image.height() };
            }
        }
        self.check(unsafe { trr_scene_set_settings(handle, &st) })?;

        let c = &scene.camera;
        let cam = TrrCamera {
            position: [c.position.x, c.position.y, c.position.z], look_at: [c.look_at.x, c.look_at.y, c.look_at.z],
            up: [c.up.x, c.up.y, c.up.z], fov: c.fov, width: c.resolution.0, height: c.resolution.1,
        };
        self.check(unsafe { trr_scene_set_camera(handle, &cam) })?;
        Ok(gpu)
    }

    fn check(&self, rc: c_int) -> anyhow::Result<()> {
        if rc == 0 { Ok(()) } else { Err(anyhow!("trayracer_hip error {rc}: {}", last_error(self.ctx))) }
    }
}

impl Drop for Backend {
    fn drop(&mut self) { unsafe { trr_destroy(self.ctx) } }
}

fn last_error(ctx: *const c_void) -> String {
    unsafe { CStr::from_ptr(trr_last_error(ctx)) }.to_string_lossy().into_owned()
}

/// 24 f32 per triangle in the declaration order of scene/triangle.rs:9-19.
fn flatten_triangles(o: &Object) -> Vec<f32> {
    let mut v = Vec::with_capacity(o.triangles.len() * 24);
    for t in &o.triangles {
        for p in [&t.a, &t.b, &t.c] { v.extend_from_slice(&[p.x, p.y, p.z]); }
        for n in [&t.a_normal, &t.b_normal, &t.c_normal] { v.extend_from_slice(&[n.x, n.y, n.z]); }
        for uv in [&t.a_uv, &t.b_uv, &t.c_uv] { v.extend_from_slice(&[uv.x, uv.y]); }
    }
    v
}

/// Device-resident flattened scene (`trr_scene`).
pub struct GpuScene<'a> { backend: &'a Backend, handle: *mut c_void }

impl Drop for GpuScene<'_> {
    fn drop(&mut self) { unsafe { trr_scene_destroy(self.handle) } }
}

struct Sinks<'a> {
    block: &'a mut dyn FnMut(&[u8], u32, u32, u32, u32),
    progress: &'a mut dyn FnMut(u32, u32),
    cancel_in: &'a AtomicBool,
    cancel_out: &'a AtomicI32,
}

extern "C" fn on_block(user: *mut c_void, xb: u32, yb: u32, bw: u32, bh: u32, rgb: *const u8) {
    let s = unsafe { &mut *(user as *mut Sinks) };
    let px = unsafe { std::slice::from_raw_parts(rgb, (bw * bh * 3) as usize) };
    (s.block)(px, xb, yb, bw, bh);
    // the reference's flag is an AtomicBool (render.rs:22); the library polls an i32 between batches
    s.cancel_out.store(i32::from(s.cancel_in.load(Ordering::Relaxed)), Ordering::Relaxed);
}

extern "C" fn on_progress(user: *mut c_void, done: u32, total: u32) {
    let s = unsafe { &mut *(user as *mut Sinks) };
    (s.progress)(done, total);
    s.cancel_out.store(i32::from(s.cancel_in.load(Ordering::Relaxed)), Ordering::Relaxed);
}

impl GpuScene<'_> {
    /// The block loop of RenderingThread::run (render.rs:103-152).  `block(rgb, x_block, y_block, w, h)` is called on this
    /// thread for every finished block, a row of the 20x20 grid at a time; `progress(done, total)` after each row.
    /// Returns Ok(false) when cancelled.
    pub fn render_blocks(&self, delta: f32, max_depth: u32, cancel: &AtomicBool, block: &mut dyn FnMut(&[u8], u32, u32, u32, u32),
                         progress: &mut dyn FnMut(u32, u32)) -> anyhow::Result<bool> {
        let flag = AtomicI32::new(i32::from(cancel.load(Ordering::Relaxed)));
        let mut sinks = Sinks { block, progress, cancel_in: cancel, cancel_out: &flag };
        let rc = unsafe {
            trr_render_blocks(self.backend.ctx, self.handle, delta, max_depth, 0, 0, flag.as_ptr().cast_const(), Some(on_block),
                              Some(on_progress), (&mut sinks as *mut Sinks).cast(), std::ptr::null_mut())
        };
        if rc == ERR_CANCELLED { return Ok(false); }
        self.backend.check(rc)?;
        Ok(true)
    }
}
