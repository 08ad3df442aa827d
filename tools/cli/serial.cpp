// serial.cpp -- ./serial <img-path> <psf-length> <psf-angle> [--out file] [--raw-out file]
// Counterpart of the reference's serial driver (BASELINE config 1; argument meaning, usage line, "Cannot read
// image", "Deblurring 3 channels took(serial)", "Total program time" and the -1 returns as at serial.cpp:9-61 of the
// reference): every channel is padded to powers of two, restored by fft_serial::wienerDeblur_myfft, cropped, then
// Lab white balance and 8 bit.  fft_serial:: in this repository runs on the GPU in the parity mode (bit-identical FFT
// arithmetic to fft/fft_serial.cpp), so the pixels are the ones ./serial computes; there is no CPU path.  The GUI
// calls (imshow / waitKey, serial.cpp:59-60) are out of scope: --out writes the picture instead.
#include "utils.hpp"
#include "fft/fft.hpp"
#include "fdr_image_io.hpp"
#include <cstdlib>
#include <iostream>
#include <string>

int main(int argc, char** argv) {
    const auto total_start = high_resolution_clock::now();
    if (argc < 4) {
        cout << "Usage: ./fft_image_restoration <img-path> <psf-length> <psf-angle>\n";
        return -1;
    }
    const string img_path = argv[1];
    const int psf_length = atoi(argv[2]);
    const double psf_angle = atof(argv[3]);
    string out_path, raw_path;
    for (int i = 4; i < argc; ++i) {
        const string a = argv[i];
        if (a == "--out" && i + 1 < argc) out_path = argv[++i];
        else if (a == "--raw-out" && i + 1 < argc) raw_path = argv[++i];
        else { cout << "Usage: ./fft_image_restoration <img-path> <psf-length> <psf-angle>\n"; return -1; }
    }

    Mat img = fdr_io::imread(img_path);
    if (img.empty()) { cout << "Cannot read image\n"; return -1; }
    img.convertTo(img, CV_32F);
    img /= 255.0;

    const Mat psf = motionBlurKernel(psf_length, psf_angle);
    const float K = 0.01f;
    vector<Mat> channels;
    split(img, channels);
    const vector<Mat> original = channels;

    const auto t_start = high_resolution_clock::now();
    for (Mat& channel : channels) {
        const Mat padded = autoPadToPowerOfTwo(channel);                         // normalisation spans the padded area ...
        const Mat restored = fft_serial::wienerDeblur_myfft(padded, psf, K);
        channel = restored(Rect(0, 0, img.cols, img.rows)).clone();             // ... and the crop comes after it
    }
    cout << "Deblurring 3 channels took(serial): " << getElapsedMs(t_start, high_resolution_clock::now()) << " ms\n";

    if (!raw_path.empty()) {
        FILE* f = fopen(raw_path.c_str(), "wb");
        if (!f) { cout << "Cannot write " << raw_path << "\n"; return -1; }
        for (const Mat& c : channels)
            for (int r = 0; r < c.rows; ++r) fwrite(c.ptr<float>(r), sizeof(float), (size_t)c.cols, f);
        fclose(f);
    }

    const float* orig[3]; const float* rest[3];
    for (int c = 0; c < 3; ++c) { orig[c] = original[c].ptr<float>(0); rest[c] = channels[c].ptr<float>(0); }
    Mat corrected_BGR(img.rows, img.cols, CV_8UC3);
    FDR_CHECK(fdr_white_balance_u8(0, orig, rest, img.rows, img.cols, img.cols, corrected_BGR.ptr<unsigned char>(0), 3 * img.cols));

    cout << "Total program time: " << getElapsedMs(total_start, high_resolution_clock::now()) << " ms\n";
    if (!out_path.empty()) {
        if (!fdr_io::imwrite(out_path, corrected_BGR)) { cout << "Cannot write " << out_path << "\n"; return -1; }
        cout << "Wrote " << out_path << "\n";
    }
    return 0;
}
