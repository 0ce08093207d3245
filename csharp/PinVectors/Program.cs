// PinVectors -- length + SHA-256 of Snappy.CompressToArray (real Snappier, Snappier/Snappy.cs:69-80) for:
//   * html[0:65536] and html[0:102400]                (the known answers SURVEY.md 8(c) / DESIGN.md 2 quote from the model),
//   * every 64 KiB window of every corpus file        (Snappier.Tests/TestData, the files of SnappyTests.cs:8-19),
//   * every corpus file whole                         (multi-fragment blocks, SnappyCompressor.cs:34-80),
// plus the facts that decide which TableEntry hash the process used (HashTable.cs:91-126): Sse42.X64 / Crc32.Arm64 support.
// Run it twice -- as is, and with DOTNET_EnableHWIntrinsic=0 -- to pin both hash variants; tests/test_snappier_pins.py
// compares every entry with oracle/snappy_oracle.c (crc32c variant when "hash" = "crc32c", mul variant otherwise).
using System.Runtime.Intrinsics.Arm;
using System.Runtime.Intrinsics.X86;
using System.Security.Cryptography;
using System.Text.Json;
using Snappier;

string dir = args.Length > 0 ? args[0] : "tests/golden/testdata";
string[] corpus = ["alice29.txt", "asyoulik.txt", "fireworks.jpeg", "geo.protodata", "html", "html_x_4", "kppkn.gtb",
                   "lcet10.txt", "paper-100k.pdf", "plrabn12.txt", "urls.10K"];

bool hwCrc = Sse42.X64.IsSupported || Crc32.Arm64.IsSupported;   // HashTable.cs:109-117 takes the CRC path exactly then
var vectors = new List<object>();

void Add(string name, byte[] data, int offset, int length)
{
    byte[] z = Snappy.CompressToArray(data.AsSpan(offset, length));
    byte[] back = Snappy.DecompressToArray(z);
    if (!back.AsSpan().SequenceEqual(data.AsSpan(offset, length))) throw new InvalidOperationException($"{name}@{offset}: round trip failed");
    vectors.Add(new { name, offset, length, compressed_length = z.Length, sha256 = Convert.ToHexString(SHA256.HashData(z)).ToLowerInvariant() });
}

foreach (string f in corpus)
{
    string path = Path.Combine(dir, f);
    byte[] data;
    if (File.Exists(path)) data = File.ReadAllBytes(path);
    else if (f == "html_x_4") { byte[] h = File.ReadAllBytes(Path.Combine(dir, "html")); data = [.. h, .. h, .. h, .. h]; }   // (= html four times: not stored twice)
    else continue;
    if (f == "html") { Add("html", data, 0, 65536); Add("html", data, 0, 102400); }
    for (int off = 0; off < data.Length; off += 65536) Add(f, data, off, Math.Min(65536, data.Length - off));
    Add(f, data, 0, data.Length);
}

var doc = new
{
    tool = "csharp/PinVectors",
    snappier_assembly = typeof(Snappy).Assembly.GetName().Version?.ToString(),
    runtime = System.Runtime.InteropServices.RuntimeInformation.FrameworkDescription,
    arch = System.Runtime.InteropServices.RuntimeInformation.ProcessArchitecture.ToString(),
    hash = hwCrc ? "crc32c" : "mul",
    sse42_x64 = Sse42.X64.IsSupported,
    crc32_arm64 = Crc32.Arm64.IsSupported,
    vectors,
};
Console.WriteLine(JsonSerializer.Serialize(doc, new JsonSerializerOptions { WriteIndented = true }));
