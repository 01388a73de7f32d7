// Golden vectors from the reference itself: PCM digests of VorbisReader.ReadSamples over a directory of .ogg files.
// Two read patterns (one large buffer; odd-sized partial reads) must give the same bytes; ClipSamples on and off are
// both recorded.  Output format, one file per input and mode:
//     <name>.pcm.sha256        : "<sha256 hex> <float count> <channels> <sample rate> clip=1"
//     <name>.noclip.pcm.sha256 : the same with ClipSamples = false
using System;
using System.IO;
using System.Security.Cryptography;
using NVorbis;

static class Program
{
    static byte[] Decode(string path, bool clip, int chunk)
    {
        using (var reader = new VorbisReader(path))
        {
            reader.ClipSamples = clip;
            var buf = new float[chunk * reader.Channels];
            using (var ms = new MemoryStream())
            {
                int n;
                while ((n = reader.ReadSamples(buf, 0, buf.Length)) > 0)
                {
                    var bytes = new byte[n * 4];
                    Buffer.BlockCopy(buf, 0, bytes, 0, bytes.Length);   // little-endian IEEE-754 on every .NET target in use
                    ms.Write(bytes, 0, bytes.Length);
                }
                return ms.ToArray();
            }
        }
    }

    static int Main(string[] args)
    {
        if (args.Length != 2) { Console.Error.WriteLine("usage: GoldenGenerator <directory with .ogg files> <output directory>"); return 2; }
        Directory.CreateDirectory(args[1]);
        foreach (var path in Directory.GetFiles(args[0], "*.ogg"))
        {
            foreach (var clip in new[] { true, false })
            {
                var a = Decode(path, clip, 4096);
                var b = Decode(path, clip, 333);
                if (a.Length != b.Length || !System.Linq.Enumerable.SequenceEqual(a, b))
                    throw new InvalidOperationException("read pattern changes the PCM of " + path);
                int channels, rate;
                using (var r = new VorbisReader(path)) { channels = r.Channels; rate = r.SampleRate; }
                string hex;
                using (var sha = SHA256.Create()) hex = BitConverter.ToString(sha.ComputeHash(a)).Replace("-", "").ToLowerInvariant();
                var name = Path.GetFileNameWithoutExtension(path) + (clip ? "" : ".noclip") + ".pcm.sha256";
                File.WriteAllText(Path.Combine(args[1], name), $"{hex} {a.Length / 4} {channels} {rate} clip={(clip ? 1 : 0)}\n");
                Console.WriteLine(name + " " + hex);
            }
        }
        return 0;
    }
}
