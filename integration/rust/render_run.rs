// UNVERIFIED (no Rust toolchain in the build image).  What changes in RenderingThread::run
// (src/raytracer/render.rs:95-193): the rayon pipeline that PRODUCES blocks (render.rs:105-152, from `Raytracer::new` to the
// closure returning `(pixels, x_block, y_block)`) is replaced by the call below; the code that CONSUMES a block -- the body
// of the `for_each_with` closure, render.rs:155-186, which paints the texture and the image and requests a repaint --
// is kept and becomes the local closure `paint` (not repeated here).
//
//     let paint = |pixels: Vec<Color32>, x_block: u32, y_block: u32| { /* render.rs:156-185, unchanged */ };
#[cfg(trayracer_hip)]
fn render_on_gpu(
    scene: &Scene,
    cancel: &AtomicBool,
    progress: &AtomicU16,
    mut paint: impl FnMut(Vec<Color32>, u32, u32),
) -> anyhow::Result<bool> {
    let gpu = hip::Backend::new(0)?;                    // one per process in a real integration (a `OnceLock`);
                                                        // hip::Backend::new_multi(&[0, 1, 2, 3, 4, 5, 6, 7]) uses a whole node
    let on_gpu = gpu.upload(scene, hip::BUILD_HOST)?;    // stands in for Raytracer::new(scene, 1e-5, 5), render.rs:105
    let mut on_block = |rgb: &[u8], x_block: u32, y_block: u32, _w: u32, _h: u32| {
        paint(rgb.chunks_exact(3).map(|p| Color32::from_rgb(p[0], p[1], p[2])).collect(), x_block, y_block);
    };
    let mut on_progress = |done: u32, total: u32| {     // same scale as render.rs:143-149
        let part = done as f32 / total.max(1) as f32;
        progress.store((part * f32::from(u16::MAX)).round() as u16, Ordering::Relaxed);
    };
    on_gpu.render_blocks(1e-5, 5, cancel, &mut on_block, &mut on_progress)
}

// In `run`, where the rayon pipeline stood:
//
//     #[cfg(trayracer_hip)]
//     let done = render_on_gpu(&self.scene, &self.cancel, &self.progress, &mut paint)
//         .map_err(|e| warn!("HIP backend: {e}"))
//         .is_ok();
//     #[cfg(not(trayracer_hip))]
//     let done = false;
//     if !done { /* the existing rayon pipeline, feeding `paint` */ }
//
// The lines after it (progress = u16::MAX, time, the log line: render.rs:188-192) stay as they are.
