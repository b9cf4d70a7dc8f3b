"""ORACLE (test infrastructure): CPU restatement of the reference training-loop bodies.

Each `*_step` executes exactly one iteration of the corresponding script's inner loop on stock torch
(forward, losses, backward, Adam), with the host RNG draws (`np.random`, `random`) in the reference's
order, and without the logging / image-saving side effects.  Line references:
  gan_step        implementations/gan/gan.py:121-161
  dcgan_step      implementations/dcgan/dcgan.py:143-183
  wgan_gp_step    implementations/wgan_gp/wgan_gp.py:119-138,146-193
  cyclegan_step   implementations/cyclegan/cyclegan.py:159-239
  pix2pix_step    implementations/pix2pix/pix2pix.py:123-172
  srgan_step      implementations/srgan/srgan.py:97-145
  dragan_step     implementations/dragan/dragan.py:144-167,176-217   (conv-critic gradient penalty)
  esrgan_step     implementations/esrgan/esrgan.py:101-174           (relativistic average GAN + warm-up)
  acgan_step      implementations/acgan/acgan.py:167-222             (label-conditioned DCGAN, auxiliary classifier)
  lsgan_step / relativistic_gan_step / ebgan_step   lsgan.py:140-180, relativistic_gan.py:126-182, ebgan.py:142-202 (8f F2 clones)
Pinned against the reference by oracle/pin_against_reference.py.
"""
import itertools
from types import SimpleNamespace

import numpy as np
import torch
from torch import autograd

from . import reference_models as M

ADAM = dict(lr=2e-4, betas=(0.5, 0.999))  # every script's argparse defaults (dcgan.py:22-24)


def _adam(params):
    return torch.optim.Adam(params, **ADAM)


def _f32(a):
    return torch.tensor(np.asarray(a), dtype=torch.float32)  # Tensor(np.float64 array) -> fp32 (dcgan.py:160)


# ------------------------------------------------------------------------------------------------ gan / dcgan
def make_gan(img_size=28, latent_dim=100, channels=1):
    shape = (channels, img_size, img_size)
    G, D = M.MlpGenerator(shape, latent_dim), M.MlpCritic(shape, sigmoid=True)
    return SimpleNamespace(G=G, D=D, opt_G=_adam(G.parameters()), opt_D=_adam(D.parameters()),
                           bce=torch.nn.BCELoss(), latent_dim=latent_dim)


def make_dcgan(img_size=32, latent_dim=100, channels=1):
    G, D = M.DcganGenerator(img_size, latent_dim, channels), M.DcganDiscriminator(img_size, channels)
    G.apply(M.init_normal_dcgan)
    D.apply(M.init_normal_dcgan)
    return SimpleNamespace(G=G, D=D, opt_G=_adam(G.parameters()), opt_D=_adam(D.parameters()),
                           bce=torch.nn.BCELoss(), latent_dim=latent_dim)


def dcgan_step(s, real_imgs, z=None):
    """One generator + one discriminator update (also the gan.py loop body)."""
    B = real_imgs.shape[0]
    valid, fake = torch.ones(B, 1), torch.zeros(B, 1)
    s.opt_G.zero_grad()
    if z is None:
        z = _f32(np.random.normal(0, 1, (B, s.latent_dim)))
    gen = s.G(z)
    g_loss = s.bce(s.D(gen), valid)
    g_loss.backward()
    s.opt_G.step()
    s.opt_D.zero_grad()
    d_loss = (s.bce(s.D(real_imgs), valid) + s.bce(s.D(gen.detach()), fake)) / 2
    d_loss.backward()
    s.opt_D.step()
    return {"g_loss": g_loss.detach(), "d_loss": d_loss.detach(), "gen_imgs": gen.detach()}


gan_step = dcgan_step


# ------------------------------------------------------------------------------------------------ wgan_gp
def make_wgan_gp(img_size=32, latent_dim=100, channels=1):
    shape = (channels, img_size, img_size)
    G, D = M.MlpGenerator(shape, latent_dim), M.MlpCritic(shape)
    return SimpleNamespace(G=G, D=D, opt_G=_adam(G.parameters()), opt_D=_adam(D.parameters()),
                           latent_dim=latent_dim, lambda_gp=10, n_critic=5)


def gradient_penalty(D, real, fake, alpha=None):
    B = real.size(0)
    if alpha is None:
        alpha = _f32(np.random.random((B, 1, 1, 1)))
    mix = (alpha * real + ((1 - alpha) * fake)).requires_grad_(True)
    out = D(mix)
    grads = autograd.grad(outputs=out, inputs=mix, grad_outputs=torch.ones(B, 1), create_graph=True,
                          retain_graph=True, only_inputs=True)[0]
    grads = grads.view(B, -1)
    return ((grads.norm(2, dim=1) - 1) ** 2).mean()


def wgan_gp_step(s, real_imgs, i, z=None, alpha=None):
    """Critic iteration i; the generator is updated when i % n_critic == 0 (same z)."""
    B = real_imgs.shape[0]
    s.opt_D.zero_grad()
    if z is None:
        z = _f32(np.random.normal(0, 1, (B, s.latent_dim)))
    fake_imgs = s.G(z)
    real_v, fake_v = s.D(real_imgs), s.D(fake_imgs)
    gp = gradient_penalty(s.D, real_imgs.data, fake_imgs.data, alpha)
    d_loss = -torch.mean(real_v) + torch.mean(fake_v) + s.lambda_gp * gp
    d_loss.backward()
    s.opt_D.step()
    s.opt_G.zero_grad()
    out = {"d_loss": d_loss.detach(), "gp": gp.detach()}
    if i % s.n_critic == 0:
        fake_imgs = s.G(z)
        g_loss = -torch.mean(s.D(fake_imgs))
        g_loss.backward()
        s.opt_G.step()
        out["g_loss"] = g_loss.detach()
    return out


# ------------------------------------------------------------------------------------------------ stargan / dualgan (8f F1)
def critic_gradient_penalty(D, real_samples, fake_samples, alpha=None):
    """stargan.py:142-161 and dualgan.py:116-135 (the same function; stargan's critic returns (out_adv, out_cls) and the
    penalty uses out_adv): alpha ~ np.random.random((B, 1, 1, 1)), gradient of the critic output w.r.t. the interpolate,
    per-sample L2 norm over all pixels."""
    if alpha is None:
        alpha = _f32(np.random.random((real_samples.size(0), 1, 1, 1)))
    interpolates = (alpha * real_samples + ((1 - alpha) * fake_samples)).requires_grad_(True)
    out = D(interpolates)
    d_interpolates = out[0] if isinstance(out, tuple) else out
    fake = _f32(np.ones(d_interpolates.shape))
    gradients = autograd.grad(outputs=d_interpolates, inputs=interpolates, grad_outputs=fake, create_graph=True,
                              retain_graph=True, only_inputs=True)[0]
    gradients = gradients.view(gradients.size(0), -1)
    return ((gradients.norm(2, dim=1) - 1) ** 2).mean()


# ------------------------------------------------------------------------------------------------ dragan (8f F1)
def make_dragan(img_size=32, latent_dim=100, channels=1):
    """dragan.py:46-99,115-120: the generator / discriminator classes are those of dcgan.py (same layer lists)."""
    s = make_dcgan(img_size, latent_dim, channels)
    s.lambda_gp = 10
    return s


def dragan_gradient_penalty(D, X, alpha=None, noise=None, lambda_gp=10):
    """dragan.py:144-167.  alpha ~ np.random.random(X.shape), noise ~ torch.rand(X.size()) when not given (the
    reference's draw order); the gradient norm is taken over dim 1 - the CHANNEL dimension of the (B, C, H, W)
    gradient - as the reference writes it."""
    if alpha is None:
        alpha = _f32(np.random.random(size=tuple(X.shape)))
    if noise is None:
        noise = torch.rand(X.size())
    interpolates = alpha * X + ((1 - alpha) * (X + 0.5 * X.std() * noise))
    interpolates = interpolates.detach().requires_grad_(True)
    d_interpolates = D(interpolates)
    fake = torch.ones(X.shape[0], 1, dtype=X.dtype)
    gradients = autograd.grad(outputs=d_interpolates, inputs=interpolates, grad_outputs=fake, create_graph=True,
                              retain_graph=True, only_inputs=True)[0]
    return lambda_gp * ((gradients.norm(2, dim=1) - 1) ** 2).mean()


def dragan_step(s, real_imgs, z=None, alpha=None, noise=None):
    """dragan.py:176-217.  Quirk kept: d_loss is computed (its two discriminator forwards update the BatchNorm running
    statistics and consume Dropout2d draws) but only gradient_penalty.backward() feeds optimizer_D.step()."""
    B = real_imgs.shape[0]
    valid, fake = torch.ones(B, 1), torch.zeros(B, 1)
    s.opt_G.zero_grad()
    if z is None:
        z = _f32(np.random.normal(0, 1, (B, s.latent_dim)))
    gen = s.G(z)
    g_loss = s.bce(s.D(gen), valid)
    g_loss.backward()
    s.opt_G.step()
    s.opt_D.zero_grad()
    d_loss = (s.bce(s.D(real_imgs), valid) + s.bce(s.D(gen.detach()), fake)) / 2
    gp = dragan_gradient_penalty(s.D, real_imgs.data, alpha, noise, s.lambda_gp)
    gp.backward()
    s.opt_D.step()
    return {"g_loss": g_loss.detach(), "d_loss": d_loss.detach(), "gp": gp.detach(), "gen_imgs": gen.detach()}


# ------------------------------------------------------------------------------------------------ cyclegan
def make_cyclegan(shape=(3, 64, 64), n_res=9):
    G_AB, G_BA = M.CycleGenerator(shape, n_res), M.CycleGenerator(shape, n_res)
    D_A, D_B = M.CycleDiscriminator(shape), M.CycleDiscriminator(shape)
    for net in (G_AB, G_BA, D_A, D_B):
        net.apply(M.init_normal_cyclegan)
    return SimpleNamespace(
        G_AB=G_AB, G_BA=G_BA, D_A=D_A, D_B=D_B,
        opt_G=_adam(itertools.chain(G_AB.parameters(), G_BA.parameters())), opt_D_A=_adam(D_A.parameters()),
        opt_D_B=_adam(D_B.parameters()), mse=torch.nn.MSELoss(), l1_cycle=torch.nn.L1Loss(),
        l1_id=torch.nn.L1Loss(), buf_A=M.ReplayBuffer(), buf_B=M.ReplayBuffer(), lambda_cyc=10.0, lambda_id=5.0)


def cyclegan_step(s, real_A, real_B):
    B = real_A.size(0)
    valid = _f32(np.ones((B, *s.D_A.output_shape)))
    fake = _f32(np.zeros((B, *s.D_A.output_shape)))
    s.G_AB.train()
    s.G_BA.train()
    s.opt_G.zero_grad()
    loss_id = (s.l1_id(s.G_BA(real_A), real_A) + s.l1_id(s.G_AB(real_B), real_B)) / 2
    fake_B = s.G_AB(real_A)
    loss_GAN_AB = s.mse(s.D_B(fake_B), valid)
    fake_A = s.G_BA(real_B)
    loss_GAN_BA = s.mse(s.D_A(fake_A), valid)
    loss_GAN = (loss_GAN_AB + loss_GAN_BA) / 2
    loss_cycle = (s.l1_cycle(s.G_BA(fake_B), real_A) + s.l1_cycle(s.G_AB(fake_A), real_B)) / 2
    loss_G = loss_GAN + s.lambda_cyc * loss_cycle + s.lambda_id * loss_id
    loss_G.backward()
    s.opt_G.step()

    s.opt_D_A.zero_grad()
    fake_A_ = s.buf_A.push_and_pop(fake_A)
    loss_D_A = (s.mse(s.D_A(real_A), valid) + s.mse(s.D_A(fake_A_.detach()), fake)) / 2
    loss_D_A.backward()
    s.opt_D_A.step()

    s.opt_D_B.zero_grad()
    fake_B_ = s.buf_B.push_and_pop(fake_B)
    loss_D_B = (s.mse(s.D_B(real_B), valid) + s.mse(s.D_B(fake_B_.detach()), fake)) / 2
    loss_D_B.backward()
    s.opt_D_B.step()
    return {"loss_G": loss_G.detach(), "loss_D": ((loss_D_A + loss_D_B) / 2).detach(), "loss_GAN": loss_GAN.detach(),
            "loss_cycle": loss_cycle.detach(), "loss_identity": loss_id.detach()}


# ------------------------------------------------------------------------------------------------ pix2pix
def make_pix2pix(img_size=256):
    G, D = M.Pix2pixGenerator(), M.Pix2pixDiscriminator()
    G.apply(M.init_normal_dcgan)
    D.apply(M.init_normal_dcgan)
    return SimpleNamespace(G=G, D=D, opt_G=_adam(G.parameters()), opt_D=_adam(D.parameters()),
                           mse=torch.nn.MSELoss(), l1=torch.nn.L1Loss(), lambda_pixel=100,
                           patch=(1, img_size // 16, img_size // 16))


def pix2pix_step(s, real_A, real_B):
    """real_A = batch["B"], real_B = batch["A"] in the script (pix2pix.py:127-128); here: condition, target."""
    B = real_A.size(0)
    valid, fake = _f32(np.ones((B, *s.patch))), _f32(np.zeros((B, *s.patch)))
    s.opt_G.zero_grad()
    fake_B = s.G(real_A)
    loss_GAN = s.mse(s.D(fake_B, real_A), valid)
    loss_pixel = s.l1(fake_B, real_B)
    loss_G = loss_GAN + s.lambda_pixel * loss_pixel
    loss_G.backward()
    s.opt_G.step()
    s.opt_D.zero_grad()
    loss_D = 0.5 * (s.mse(s.D(real_B, real_A), valid) + s.mse(s.D(fake_B.detach(), real_A), fake))
    loss_D.backward()
    s.opt_D.step()
    return {"loss_G": loss_G.detach(), "loss_D": loss_D.detach(), "loss_pixel": loss_pixel.detach(),
            "loss_GAN": loss_GAN.detach()}


# ------------------------------------------------------------------------------------------------ srgan
def make_srgan(hr_shape=(96, 96), n_res=16):
    G = M.SrganGenerator(n_residual_blocks=n_res)
    D = M.SrganDiscriminator(input_shape=(3, *hr_shape))
    V = M.SrganFeatureExtractor()
    V.eval()
    return SimpleNamespace(G=G, D=D, V=V, opt_G=_adam(G.parameters()), opt_D=_adam(D.parameters()),
                           mse=torch.nn.MSELoss(), l1=torch.nn.L1Loss())


def srgan_step(s, imgs_lr, imgs_hr):
    B = imgs_lr.size(0)
    valid = _f32(np.ones((B, *s.D.output_shape)))
    fake = _f32(np.zeros((B, *s.D.output_shape)))
    s.opt_G.zero_grad()
    gen_hr = s.G(imgs_lr)
    loss_GAN = s.mse(s.D(gen_hr), valid)
    loss_content = s.l1(s.V(gen_hr), s.V(imgs_hr).detach())
    loss_G = loss_content + 1e-3 * loss_GAN
    loss_G.backward()
    s.opt_G.step()
    s.opt_D.zero_grad()
    loss_D = (s.mse(s.D(imgs_hr), valid) + s.mse(s.D(gen_hr.detach()), fake)) / 2
    loss_D.backward()
    s.opt_D.step()
    return {"loss_G": loss_G.detach(), "loss_D": loss_D.detach(), "loss_content": loss_content.detach(),
            "loss_GAN": loss_GAN.detach()}


# ------------------------------------------------------------------------------------------------ esrgan
def make_esrgan(hr_shape=(64, 64), n_res=23, filters=64):
    """esrgan.py:60-83: GeneratorRRDB(channels, 64, residual_blocks), Discriminator, VGG19[:35] in eval mode,
    Adam(lr 2e-4, betas (0.9, 0.999)) - this script's --b1 default is 0.9 (esrgan.py:40)."""
    G = M.EsrganGenerator(3, filters=filters, num_res_blocks=n_res)
    D = M.EsrganDiscriminator(input_shape=(3, *hr_shape))
    V = M.EsrganFeatureExtractor()
    V.eval()
    adam = lambda p: torch.optim.Adam(p, lr=2e-4, betas=(0.9, 0.999))  # noqa: E731
    return SimpleNamespace(G=G, D=D, V=V, opt_G=adam(G.parameters()), opt_D=adam(D.parameters()),
                           bce_logits=torch.nn.BCEWithLogitsLoss(), l1_content=torch.nn.L1Loss(), l1_pixel=torch.nn.L1Loss(),
                           warmup_batches=500, lambda_adv=5e-3, lambda_pixel=1e-2)


def esrgan_step(s, imgs_lr, imgs_hr, batches_done):
    """esrgan.py:101-174.  During warm-up (batches_done < warmup_batches) only the pixel loss trains G (esrgan.py:123-131)."""
    B = imgs_lr.size(0)
    valid = _f32(np.ones((B, *s.D.output_shape)))
    fake = _f32(np.zeros((B, *s.D.output_shape)))
    s.opt_G.zero_grad()
    gen_hr = s.G(imgs_lr)
    loss_pixel = s.l1_pixel(gen_hr, imgs_hr)
    if batches_done < s.warmup_batches:
        loss_pixel.backward()
        s.opt_G.step()
        return {"loss_pixel": loss_pixel.detach()}
    pred_real = s.D(imgs_hr).detach()
    pred_fake = s.D(gen_hr)
    loss_GAN = s.bce_logits(pred_fake - pred_real.mean(0, keepdim=True), valid)
    gen_features = s.V(gen_hr)
    real_features = s.V(imgs_hr).detach()
    loss_content = s.l1_content(gen_features, real_features)
    loss_G = loss_content + s.lambda_adv * loss_GAN + s.lambda_pixel * loss_pixel
    loss_G.backward()
    s.opt_G.step()
    s.opt_D.zero_grad()
    pred_real = s.D(imgs_hr)
    pred_fake = s.D(gen_hr.detach())
    loss_real = s.bce_logits(pred_real - pred_fake.mean(0, keepdim=True), valid)
    loss_fake = s.bce_logits(pred_fake - pred_real.mean(0, keepdim=True), fake)
    loss_D = (loss_real + loss_fake) / 2
    loss_D.backward()
    s.opt_D.step()
    return {"loss_G": loss_G.detach(), "loss_D": loss_D.detach(), "loss_content": loss_content.detach(),
            "loss_GAN": loss_GAN.detach(), "loss_pixel": loss_pixel.detach()}


# ------------------------------------------------------------------------------------------------ acgan (8f F2)
def make_acgan(img_size=32, latent_dim=100, channels=1, n_classes=10):
    """acgan.py:110-125,151-152: BCELoss + CrossEntropyLoss, weights_init_normal on both networks (it touches Conv and
    BatchNorm2d layers only: the Embedding and Linear layers keep torch's default init)."""
    G, D = M.AcganGenerator(img_size, latent_dim, channels, n_classes), M.AcganDiscriminator(img_size, channels, n_classes)
    G.apply(M.init_normal_dcgan)
    D.apply(M.init_normal_dcgan)
    return SimpleNamespace(G=G, D=D, opt_G=_adam(G.parameters()), opt_D=_adam(D.parameters()), bce=torch.nn.BCELoss(),
                           ce=torch.nn.CrossEntropyLoss(), latent_dim=latent_dim, n_classes=n_classes)


def acgan_step(s, real_imgs, labels, z=None, gen_labels=None):
    """acgan.py:167-222.  Quirk kept: CrossEntropyLoss is fed the Softmax OUTPUT of the class head (a log-softmax of
    probabilities).  Host draws when not given, in the reference's order: z ~ np.random.normal, gen_labels ~ np.random.randint.
    The class accuracy print-out (acgan.py:216-219) is logging and is not restated."""
    B = real_imgs.shape[0]
    valid, fake = torch.ones(B, 1), torch.zeros(B, 1)
    s.opt_G.zero_grad()
    if z is None:
        z = _f32(np.random.normal(0, 1, (B, s.latent_dim)))
    if gen_labels is None:
        gen_labels = torch.tensor(np.random.randint(0, s.n_classes, B), dtype=torch.long)
    gen_imgs = s.G(z, gen_labels)
    validity, pred_label = s.D(gen_imgs)
    g_loss = 0.5 * (s.bce(validity, valid) + s.ce(pred_label, gen_labels))
    g_loss.backward()
    s.opt_G.step()
    s.opt_D.zero_grad()
    real_pred, real_aux = s.D(real_imgs)
    d_real_loss = (s.bce(real_pred, valid) + s.ce(real_aux, labels)) / 2
    fake_pred, fake_aux = s.D(gen_imgs.detach())
    d_fake_loss = (s.bce(fake_pred, fake) + s.ce(fake_aux, gen_labels)) / 2
    d_loss = (d_real_loss + d_fake_loss) / 2
    d_loss.backward()
    s.opt_D.step()
    return {"g_loss": g_loss.detach(), "d_loss": d_loss.detach(), "gen_imgs": gen_imgs.detach()}



# ------------------------------------------------------------------------------------------------ DCGAN-block clones (8f F2)
def make_clone(name, img_size=32):
    """Networks of lsgan / sgan / infogan / relativistic_gan / cogan / began / ebgan as their scripts build them (the
    script's weights_init_normal where it applies one; relativistic_gan.py keeps torch's default init) + Adam(2e-4, (.5, .999))."""
    G, D, init = M.clone_models(name, img_size)
    if init is not None:
        G.apply(init)
        D.apply(init)
    return SimpleNamespace(G=G, D=D, opt_G=_adam(G.parameters()), opt_D=_adam(D.parameters()), name=name,
                           latent_dim=G.l1[0].in_features if hasattr(G, "l1") else 100)


def lsgan_step(s, real_imgs, z):
    """lsgan.py:140-180: the dcgan.py loop with MSELoss on an unbounded validity and 0.5 * (real + fake)."""
    mse = torch.nn.MSELoss()
    B = real_imgs.shape[0]
    valid, fake = torch.ones(B, 1), torch.zeros(B, 1)
    s.opt_G.zero_grad()
    gen = s.G(z)
    g_loss = mse(s.D(gen), valid)
    g_loss.backward()
    s.opt_G.step()
    s.opt_D.zero_grad()
    d_loss = 0.5 * (mse(s.D(real_imgs), valid) + mse(s.D(gen.detach()), fake))
    d_loss.backward()
    s.opt_D.step()
    return {"g_loss": g_loss.detach(), "d_loss": d_loss.detach(), "gen_imgs": gen.detach()}


def relativistic_gan_step(s, real_imgs, z, rel_avg_gan=False):
    """relativistic_gan.py:126-182.  Quirk kept (relativistic_gan.py:148-157): the generator step evaluates D(real) and
    D(gen), forms the relativistic loss - and then OVERWRITES it with the plain BCEWithLogits(D(gen), valid) of a third
    discriminator forward.  The two discarded forwards still draw Dropout2d masks and move the BatchNorm running statistics;
    the discriminator step (relativistic_gan.py:166-182) is the real relativistic loss."""
    bce = torch.nn.BCEWithLogitsLoss()
    B = real_imgs.shape[0]
    valid, fake = torch.ones(B, 1), torch.zeros(B, 1)
    s.opt_G.zero_grad()
    gen = s.G(z)
    real_pred = s.D(real_imgs).detach()
    fake_pred = s.D(gen)
    if rel_avg_gan:
        g_loss = bce(fake_pred - real_pred.mean(0, keepdim=True), valid)
    else:
        g_loss = bce(fake_pred - real_pred, valid)
    g_loss = bce(s.D(gen), valid)
    g_loss.backward()
    s.opt_G.step()
    s.opt_D.zero_grad()
    real_pred = s.D(real_imgs)
    fake_pred = s.D(gen.detach())
    if rel_avg_gan:
        real_loss = bce(real_pred - fake_pred.mean(0, keepdim=True), valid)
        fake_loss = bce(fake_pred - real_pred.mean(0, keepdim=True), fake)
    else:
        real_loss = bce(real_pred - fake_pred, valid)
        fake_loss = bce(fake_pred - real_pred, fake)
    d_loss = (real_loss + fake_loss) / 2
    d_loss.backward()
    s.opt_D.step()
    return {"g_loss": g_loss.detach(), "d_loss": d_loss.detach(), "gen_imgs": gen.detach()}


def pullaway_loss(embeddings):
    """ebgan.py:142-148: mean off-diagonal cosine similarity of the batch's embeddings."""
    norm = torch.sqrt(torch.sum(embeddings ** 2, -1, keepdim=True))
    normalized_emb = embeddings / norm
    similarity = torch.matmul(normalized_emb, normalized_emb.transpose(1, 0))
    batch_size = embeddings.size(0)
    return (torch.sum(similarity) - batch_size) / (batch_size * (batch_size - 1))


def ebgan_step(s, real_imgs, z, opt_batch_size=64, lambda_pt=0.1):
    """ebgan.py:159-202: auto-encoder discriminator; G: reconstruction MSE + 0.1 * pull-away term; D: real reconstruction
    + hinge max(0, margin - fake reconstruction) with margin = max(1, batch_size / 64) (ebgan.py:157) decided on the HOST
    from .item() (ebgan.py:198)."""
    mse = torch.nn.MSELoss()
    margin = max(1, opt_batch_size / 64.0)
    s.opt_G.zero_grad()
    gen = s.G(z)
    recon, emb = s.D(gen)
    g_loss = mse(recon, gen.detach()) + lambda_pt * pullaway_loss(emb)
    g_loss.backward()
    s.opt_G.step()
    s.opt_D.zero_grad()
    real_recon, _ = s.D(real_imgs)
    fake_recon, _ = s.D(gen.detach())
    d_loss_real = mse(real_recon, real_imgs)
    d_loss_fake = mse(fake_recon, gen.detach())
    d_loss = d_loss_real
    if (margin - d_loss_fake.data).item() > 0:
        d_loss = d_loss + (margin - d_loss_fake)
    d_loss.backward()
    s.opt_D.step()
    return {"g_loss": g_loss.detach(), "d_loss": d_loss.detach(), "gen_imgs": gen.detach()}
