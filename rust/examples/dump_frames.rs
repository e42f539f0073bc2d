//! Pin kit, reference side: render fixed frames with the UNMODIFIED thomasantony/splat (its own euc rasteriser)
//! and write color.raw() as little-endian u32 files.  Copy to `examples/dump_frames.rs` of a checkout of the
//! reference and run `cargo run --release --example dump_frames -- OUT_DIR [c1.ply]`.
//! UNTESTED here (no cargo in the authoring image).  tools/pin_euc.py compares the dumps with oracle/.
use std::io::Write;

use euc::Buffer2d;
use nalgebra::Vector3;
use splat::camera::Camera;
use splat::gaussians::{self, Gaussian, GaussianList};
use splat::pipelines::{GaussianSplatPipeline01, GaussianSplatPipeline02};

fn dump(dir: &str, name: &str, color: &euc::Buffer<u32, 2>) {
    let mut f = std::fs::File::create(format!("{}/{}", dir, name)).unwrap();
    for px in color.raw() {
        f.write_all(&px.to_le_bytes()).unwrap();
    }
    println!("wrote {}/{}", dir, name);
}

fn with_cov3d(mut g: Vec<Gaussian>) -> Vec<Gaussian> {
    for x in g.iter_mut() {
        x.compute_cov3d(); // src/main.rs:24-26
    }
    g
}

fn main() {
    let args: Vec<String> = std::env::args().collect();
    let dir = args.get(1).map(|s| s.as_str()).unwrap_or(".");

    // (1) the default binary's frame on the naive scene: 800 x 600, (0,0,5), pose updated (src/main.rs:9-34, 69-78)
    {
        let (w, h) = (800usize, 600usize);
        let mut color: euc::Buffer<u32, 2> = Buffer2d::fill([w, h], 0);
        let mut camera = Camera::new(h as f32, w as f32, Some(Vector3::new(0.0, 0.0, 5.0)));
        camera.update_camera_pose();
        let p = GaussianSplatPipeline01 { gaussians: with_cov3d(gaussians::naive_gaussians()), camera };
        p.render_to_buffer(&mut color);
        dump(dir, "naive_800x600_p01.raw", &color);
    }
    // (2) src/bin/01_naive_gaussian.rs:21-32 as it is: the matrices are never computed (identity)
    {
        let (w, h) = (1280usize, 720usize);
        let mut color: euc::Buffer<u32, 2> = Buffer2d::fill([w, h], 0);
        let camera = Camera::new(h as f32, w as f32, Some(Vector3::new(-0.57651054, 2.99040512, -0.03924271)));
        let p = GaussianSplatPipeline01 { gaussians: with_cov3d(gaussians::naive_gaussians()), camera };
        p.render_to_buffer(&mut color);
        dump(dir, "naive_1280x720_p01_identity.raw", &color);
    }
    // (3) the SoA pipeline (lowpass 0.3): 1280 x 720, (0,0,3), pose updated
    {
        let (w, h) = (1280usize, 720usize);
        let mut color: euc::Buffer<u32, 2> = Buffer2d::fill([w, h], 0);
        let mut camera = Camera::new(h as f32, w as f32, Some(Vector3::new(0.0, 0.0, 3.0)));
        camera.update_camera_pose();
        let p = GaussianSplatPipeline02 { gaussians: GaussianList::from_vec(gaussians::naive_gaussians()), camera };
        p.render_to_buffer(&mut color);
        dump(dir, "naive_1280x720_p02.raw", &color);
    }
    // (4) C1: the 10k synthetic PLY (tools/pin_euc.py --write-c1 c1.ply), 256 x 256, (0,0,5)
    if let Some(ply) = args.get(2) {
        let (w, h) = (256usize, 256usize);
        let mut color: euc::Buffer<u32, 2> = Buffer2d::fill([w, h], 0);
        let mut camera = Camera::new(h as f32, w as f32, Some(Vector3::new(0.0, 0.0, 5.0)));
        camera.update_camera_pose();
        let p = GaussianSplatPipeline01 { gaussians: with_cov3d(gaussians::load_from_ply(ply)), camera };
        p.render_to_buffer(&mut color);
        dump(dir, "c1_256x256_p01.raw", &color);
    }
}
