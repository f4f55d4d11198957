// UNVERIFIED (no Rust toolchain in the build image).  Body of RenderingThread::run (src/raytracer/render.rs:95-193) with
// the rayon block loop (render.rs:105-152) replaced by the HIP backend; the sink (render.rs:155-186) is unchanged in
// what it does.  Falls back to the existing rayon path when no HIP device can be opened.
fn run(self) {
    let start = Instant::now();
    let (width, height) = self.image.lock().dimensions();

    let rendered = (|| -> anyhow::Result<()> {
        let gpu = hip::Backend::new(0)?;                                   // once per process in a real integration
        let scene = gpu.upload(&self.scene, hip::BUILD_HOST)?;             // replaces Raytracer::new(self.scene, 1e-5, 5)
        let texture = &self.texture;
        let image = &self.image;
        let ctx = &self.ctx;
        let progress = &self.progress;
        let mut on_block = |rgb: &[u8], x_block: u32, y_block: u32, bw: u32, bh: u32| {
            let pixels = rgb.chunks_exact(3).map(|p| Color32::from_rgb(p[0], p[1], p[2])).collect::<Vec<_>>();
            texture.clone().set_partial(
                [(x_block * bw) as usize, (y_block * bh) as usize],
                ImageData::Color(Arc::new(ColorImage { size: [bw as usize, bh as usize], pixels })),
                TextureOptions::default(),
            );
            let mut image = image.lock();
            for y in 0..bh {
                for x in 0..bw {
                    let i = ((x + y * bw) * 3) as usize;
                    image.put_pixel(x_block * bw + x, y_block * bh + y, image::Rgb([rgb[i], rgb[i + 1], rgb[i + 2]]));
                }
            }
            ctx.request_repaint();
        };
        let mut on_progress = |done: u32, total: u32| {                     // render.rs:143-149
            progress.store((done as f32 / total as f32 * f32::from(u16::MAX)).round() as u16, Ordering::Relaxed);
        };
        scene.render_blocks(1e-5, 5, &self.cancel, &mut on_block, &mut on_progress)?;
        Ok(())
    })();
    if let Err(e) = rendered {
        warn!("HIP backend unavailable ({e}); rendering on the CPU");
        // ... the existing rayon block loop of render.rs:103-186 ...
    }
    debug_assert!(width >= 20 && height >= 20);                            // the backend returns an error below 20 px

    self.progress.store(u16::MAX, Ordering::Relaxed);                       // render.rs:188
    self.time.store(start.elapsed().as_millis() as u32, Ordering::Relaxed); // render.rs:189-190
    info!("rendering finished: {:?}", start.elapsed());
}
